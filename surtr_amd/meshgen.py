"""Synthetic closed triangle meshes for the headless harness, tests and bench.

These replace the reference's OBJ assets (Resources/Models/*.obj are reference
content and are not copied).  Conventions follow what the reference has after
Surtr::LoadModelData (Src/Surtr.cpp:2683-2727): welded vertices, triangles
wound counter-clockwise seen from outside (x negated + winding flipped by
assimp cancel out), model scale applied.
"""
import numpy as np


def cube(scale=3.0):
    """cfg1: 8 vertices / 12 triangles, the quads of Resources/Models/cube.obj
    split (a,b,c),(a,c,d); extent = 2*scale (Src/Surtr.cpp:1403 uses scale 3)."""
    v = np.array([[-1, -1, 1], [-1, 1, 1], [-1, -1, -1], [-1, 1, -1],
                  [1, -1, 1], [1, 1, 1], [1, -1, -1], [1, 1, -1]], np.float32) * np.float32(scale)
    quads = [(0, 1, 3, 2), (2, 3, 7, 6), (6, 7, 5, 4), (4, 5, 1, 0), (2, 6, 4, 0), (7, 3, 1, 5)]
    tris = []
    for a, b, c, d in quads:
        tris += [(a, b, c), (a, c, d)]
    tris = np.array(tris, np.int32)
    return _outward(v, tris)


def _outward(v, tris):
    """Flip the winding if the signed volume is negative."""
    p = v.astype(np.float64)
    vol = np.einsum("ij,ij->i", p[tris[:, 0]], np.cross(p[tris[:, 1]], p[tris[:, 2]])).sum() / 6.0
    if vol < 0:
        tris = tris[:, ::-1].copy()
    return v.astype(np.float32), tris.astype(np.int32)


def icosphere(level):
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t),
         (0, -1, -t), (0, 1, -t), (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    v = [np.array(p, np.float64) / np.linalg.norm(p) for p in v]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2),
         (10, 7, 6), (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5),
         (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    for _ in range(level):
        cache = {}
        nf = []

        def mid(a, b):
            key = (min(a, b), max(a, b))
            if key not in cache:
                m = v[a] + v[b]
                v.append(m / np.linalg.norm(m))
                cache[key] = len(v) - 1
            return cache[key]

        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = nf
    return np.array(v, np.float64), np.array(f, np.int32)


def blob(level=4, scale=1.0):
    """cfg2/cfg3 stand-in for the low-poly closed bunny (2 503 v / 5 002 tri):
    a bumpy, non-convex icosphere with 2 562 v / 5 120 tri at level 4."""
    v, f = icosphere(level)
    x, y, z = v[:, 0], v[:, 1], v[:, 2]
    r = 1.0 + 0.28 * np.sin(3.0 * x + 0.5) * np.cos(2.0 * y - 0.3) + 0.22 * np.sin(4.0 * z + 1.1) * np.cos(3.0 * x * y)
    r += 0.35 * np.exp(-8.0 * ((x - 0.5) ** 2 + (y - 0.6) ** 2 + (z - 0.62) ** 2))      # an "ear"
    p = v * r[:, None] * np.array([1.0, 0.8, 0.7]) * scale
    return _outward(p.astype(np.float32), f)


def urchin(level=4, scale=1.0, spikes=14, length=1.6, width=0.16):
    """Closed genus-0 mesh with deep lobes for the island / non-convex ear-clipping paths of cfg2/cfg3 (the bunny's ears in
    the large): an icosphere with `spikes` long thin spikes along fixed directions, bent sideways so that the surface is
    not star-shaped.  A Voronoi cell between two spikes holds a piece of each: several islands per cell."""
    v, f = icosphere(level)
    # spike axes: a fixed spherical Fibonacci set (deterministic, no RNG)
    k = np.arange(spikes, dtype=np.float64) + 0.5
    phi = np.arccos(1.0 - 2.0 * k / spikes)
    theta = np.pi * (1.0 + 5.0 ** 0.5) * k
    axes = np.stack([np.cos(theta) * np.sin(phi), np.sin(theta) * np.sin(phi), np.cos(phi)], 1)
    cosang = np.clip(v @ axes.T, -1.0, 1.0)
    ang = np.arccos(cosang)
    bump = np.exp(-(ang / width) ** 2)                        # one narrow bump per axis
    r = 0.6 + length * bump.max(1)
    p = v * r[:, None]
    # bend every spike sideways in proportion to the square of the height above the core (overhangs: not star-shaped)
    which = bump.argmax(1)
    side = np.cross(axes[which], np.roll(axes, 1, axis=0)[which])
    side /= np.maximum(np.linalg.norm(side, axis=1, keepdims=True), 1e-12)
    h = np.maximum(r - 0.6, 0.0)
    p = p + side * (0.35 * h * h)[:, None]
    return _outward((p * scale).astype(np.float32), f)


def bumpy_torus(nu=250, nv=200, R=1.0, r0=0.35):
    """cfg4: closed bumpy torus grid nu x nv -> nu*nv vertices,
    2*nu*nv triangles, valence 6 everywhere.  250 x 200 = 50 000 v / 100 000 tri."""
    u = np.arange(nu, dtype=np.float64) * (2.0 * np.pi / nu)
    w = np.arange(nv, dtype=np.float64) * (2.0 * np.pi / nv)
    U, W = np.meshgrid(u, w, indexing="ij")
    r = r0 * (1.0 + 0.25 * np.sin(5.0 * U) * np.cos(3.0 * W))
    p = np.stack([(R + r * np.cos(W)) * np.cos(U), (R + r * np.cos(W)) * np.sin(U), r * np.sin(W)], axis=-1)
    idx = np.arange(nu * nv, dtype=np.int64).reshape(nu, nv)
    a = idx
    b = np.roll(idx, -1, axis=0)
    c = np.roll(np.roll(idx, -1, axis=0), -1, axis=1)
    d = np.roll(idx, -1, axis=1)
    tris = np.concatenate([np.stack([a, b, c], -1).reshape(-1, 3), np.stack([a, c, d], -1).reshape(-1, 3)])
    return _outward(p.reshape(-1, 3).astype(np.float32), tris.astype(np.int32))


def signed_volume(v, tris):
    p = v.astype(np.float64)
    return float(np.einsum("ij,ij->i", p[tris[:, 0]], np.cross(p[tris[:, 1]], p[tris[:, 2]])).sum() / 6.0)
