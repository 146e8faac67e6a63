"""The BASELINE.json configurations as synthetic scenes (inputs of one fracture event).

Everything here is product-side harness code: meshes from meshgen, neighbour
rings and Voronoi cells through the C ABI host helpers, seeds restated from
Surtr::GenerateVoronoi / GenerateFracturePattern (Src/Surtr.cpp:1984-2001,
2072-2096).  Nothing here imports the oracle.
"""
import numpy as np

from . import engine, meshgen

SEED = 46354  # FractureArgs::Seed, Inc/Surtr.h:89-110


def _canonical(raw):
    """std::generate_canonical<double,53> over mt19937 (two 32-bit draws per value, libstdc++)."""
    lo = raw[0::2].astype(np.float64)
    hi = raw[1::2].astype(np.float64)
    r = (lo + hi * 4294967296.0) / 18446744073709551616.0
    return np.minimum(r, np.nextafter(1.0, 0.0))


def _mt(seed):
    bg = np.random.MT19937()
    bg._legacy_seeding(seed)      # init_genrand(seed) == std::mt19937(seed)
    return bg


def uniform_seeds(n, seed=SEED):
    """Surtr::GenerateVoronoi(int): uniform(-0.5,0.5)^3 in x,y,z draw order, narrowed to float."""
    raw = _mt(seed).random_raw(6 * n)
    r = _canonical(raw)
    v = r * 1.0 + (-0.5)
    return v.reshape(n, 3).astype(np.float32).astype(np.float64)


def pattern_seeds(n, mean, seed=SEED):
    """Surtr::GenerateFracturePattern: exponential length (clamped) x normalised uniform(-1,1)^3 direction."""
    raw = _mt(seed).random_raw(8 * n)
    r = _canonical(raw).reshape(n, 4)
    length = np.maximum(np.minimum(-np.log(1.0 - r[:, 0]) / (1.0 / mean), 0.5), 1e-12)
    d = (r[:, 1:4] * 2.0 + (-1.0)).astype(np.float32)
    t = d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]
    ln = np.sqrt((t + d[:, 2] * d[:, 2]).astype(np.float32)).astype(np.float32)
    d = (d / ln[:, None]).astype(np.float32)
    return (d * length.astype(np.float32)[:, None]).astype(np.float32).astype(np.float64)


def box_solid(extent, center, factor=2.0):
    """Poly::GetBB scaled by the AABB extent and by 2, translated to the AABB centre
    (PrepareFracture step 5, Src/Surtr.cpp:1779-1782); ach_convex() applies the k-DOP clip of step 6."""
    p = np.array([[-.5, -.5, -.5], [.5, -.5, -.5], [.5, .5, -.5], [-.5, .5, -.5],
                  [-.5, -.5, .5], [.5, -.5, .5], [.5, .5, .5], [-.5, .5, .5]], np.float32)
    nb = np.array([[1, 4, 3], [5, 0, 2], [3, 6, 1], [7, 2, 0], [5, 7, 0], [1, 6, 4], [5, 2, 7], [4, 6, 3]], np.int32)
    pos = (p * np.asarray(extent, np.float32)) * np.float32(factor) + np.asarray(center, np.float32)
    return {"pos": pos.astype(np.float32), "off": np.arange(0, 25, 3, dtype=np.uint32), "nbr": nb.reshape(-1)}


def mesh_scene(verts, tris, eng=None):
    """The piece alone (no pattern): neighbour rings (on the device when an Engine is given), bounding box, the 2x box Convex."""
    mesh = eng.neighbors_from_mesh(verts, tris)[0] if eng is not None else engine.neighbors_from_mesh(verts, tris)
    lo, hi = verts.min(0), verts.max(0)
    extent = (hi - lo).astype(np.float32)
    center = ((hi.astype(np.float64) + lo.astype(np.float64)) / 2.0).astype(np.float32)
    return {"mesh": mesh, "convex": box_solid(extent, center), "tris": tris, "scale": extent, "translate": center}


def make_scene(verts, tris, n_cells, seeds=None, eng=None):
    """eng: an Engine -> the Voronoi cells are built on the device (surtr_build_cells, which also leaves them installed as
    that engine's pattern); None -> the host builder (CPU tier, no GPU needed).  Same cells either way."""
    mesh = engine.neighbors_from_mesh(verts, tris)
    lo, hi = verts.min(0), verts.max(0)
    extent = (hi - lo).astype(np.float32)
    center = ((hi.astype(np.float64) + lo.astype(np.float64)) / 2.0).astype(np.float32)
    if seeds is None:
        seeds = uniform_seeds(n_cells)
    if eng is not None:
        eng.build_cells(seeds)
        cells = eng.download_cells()
        face_off, v012 = cells["cell_face_off"], cells["v012"]
    else:
        cells = engine.voronoi_cells(seeds)
        face_off, v012 = engine.pattern_from_cells(cells)
    return {"mesh": mesh, "convex": box_solid(extent, center), "tris": tris, "seeds": seeds, "cells": cells,
            "face_off": face_off, "v012": v012, "scale": extent, "translate": center, "n_cells": n_cells}


def cube_scene(n_cells=8):
    v, t = meshgen.cube()
    return make_scene(v, t, n_cells)


def blob_scene(n_cells=64):
    v, t = meshgen.blob(scale=70.0)
    return make_scene(v, t, n_cells)


def urchin_scene(n_cells=64):
    """Deep-lobed stand-in for the bunny's ears (see meshgen.urchin): cells that hold several islands -- 29 % of the
    non-empty cells at 64 cells (14 long spikes, 2 562 vertices), 48 % at 1 024 cells (160 thin spikes, 10 242 vertices)."""
    if n_cells <= 256:
        v, t = meshgen.urchin(scale=70.0)
    else:
        v, t = meshgen.urchin(level=5, scale=70.0, spikes=160, length=1.0, width=0.06)
    return make_scene(v, t, n_cells)


def torus_scene(n_cells=4096, eng=None):
    v, t = meshgen.bumpy_torus()
    return make_scene(v, t, n_cells, eng=eng)


def fragments_as_pieces(frags):
    """Turns the fragment set of an event (engine.download / merge_fragments) into input pieces."""
    meshes, convexes = [], []
    for k in range(frags["frag_ids"].shape[0]):
        for pre, out in (("mesh", meshes), ("conv", convexes)):
            vo, no = frags[pre + "_vert_off"], frags[pre + "_nbr_off"]
            a, b = int(vo[k]), int(vo[k + 1])
            out.append({"pos": frags[pre + "_pos"][a:b].copy(), "off": (no[a:b + 1] - no[a]).astype(np.uint32),
                        "nbr": frags[pre + "_nbr"][int(no[a]):int(no[b])].copy()})
    return meshes, convexes


def refracture_scene(meshes, convexes, cells_per_piece, seed=SEED):
    """BASELINE configs[4] / SURVEY 8(d) cfg5: every first-level fragment gets `cells_per_piece` Voronoi cells drawn with
    seed + fragment index and placed in that fragment's bounding box.  Returns the concatenated pattern, the
    per-group placement and the fragment-major (cell, piece) pair list."""
    face_off = [np.zeros(1, np.uint32)]
    v012, scales, shifts, pair_cell, pair_piece = [], [], [], [], []
    base_face, base_cell = 0, 0
    for p, m in enumerate(meshes):
        lo, hi = m["pos"].min(0), m["pos"].max(0)
        scales.append((hi - lo).astype(np.float32))
        shifts.append(((hi.astype(np.float64) + lo.astype(np.float64)) / 2.0).astype(np.float32))
        cells = engine.voronoi_cells(uniform_seeds(cells_per_piece, seed + p))
        fo, v = engine.pattern_from_cells(cells)
        face_off.append((fo[1:].astype(np.int64) + base_face).astype(np.uint32))
        v012.append(v)
        base_face += int(fo[-1])
        pair_cell += list(range(base_cell, base_cell + cells_per_piece))
        pair_piece += [p] * cells_per_piece
        base_cell += cells_per_piece
    return {"face_off": np.concatenate(face_off), "v012": np.concatenate(v012), "scales": np.array(scales, np.float32),
            "shifts": np.array(shifts, np.float32), "group_cell_off": np.arange(0, base_cell + 1, cells_per_piece, dtype=np.uint32),
            "pair_cell": np.array(pair_cell, np.uint32), "pair_piece": np.array(pair_piece, np.uint32), "n_cells": base_cell}


ICH_POINT_LIMIT = 20       # FractureArgs::ICHIncludePointLimit, Inc/Surtr.h:89-110
ACH_PLANE_GAP_INV = 2000.0  # FractureArgs::ACHPlaneGapInverse


def ach_convex(eng, verts):
    """PrepareFracture steps 1-6 (Src/Surtr.cpp:1750-1785): ICH(20) face normals -> k-DOP slabs pushed out by
    MaxAxisScale/2000 -> the 2x bounding box clipped by all of them (on the GPU, through surtr_clip_polyhedron)."""
    v = np.ascontiguousarray(verts, np.float32)
    lo, hi = v.min(0), v.max(0)
    extent = (hi - lo).astype(np.float32)
    center = ((hi.astype(np.float64) + lo.astype(np.float64)) / 2.0).astype(np.float32)
    normals = engine.hull_normals(v, ICH_POINT_LIMIT)
    max_axis = float(max(float(hi[0]) - float(lo[0]), float(hi[1]) - float(lo[1]), float(hi[2]) - float(lo[2])))
    planes = engine.kdop_ach_planes(v, normals, max_axis, ACH_PLANE_GAP_INV)
    return eng.clip_polyhedron(box_solid(extent, center), planes), planes


def prepare_fracture(eng, verts, tris, n_cells=64, seed=SEED, flags=engine.EVT_REFIT | engine.EVT_RENDER):
    """Surtr::PrepareFracture end to end (Src/Surtr.cpp:1747-1827) from a raw triangle mesh: ICH(20) normals ->
    k-DOP slabs -> ACH = clipped 2x box (steps 1-6), neighbour rings of the mesh (7), `n_cells` Voronoi cells scaled by
    the AABB extent and centred (8), one fracture event of the single piece (Mesh, ACH) with refit and extraction (10).
    Returns (scene, counts, fragments); scenes.fragments_as_pieces(fragments) gives the pieces of the initial compound
    in the reference's order (bind 0 is empty, then one bind per cell: cell-major = fragment order)."""
    v = np.ascontiguousarray(verts, np.float32)
    sc = make_scene(v, np.ascontiguousarray(tris, np.int32), n_cells, seeds=uniform_seeds(n_cells, seed))
    sc["convex"], sc["ach_planes"] = ach_convex(eng, v)
    eng.upload_pieces([sc["mesh"]], [sc["convex"]])
    eng.upload_pattern(sc["face_off"], sc["v012"])
    eng.place_cells(sc["scale"], sc["translate"])
    counts = eng.fracture_event(0, n_cells, flags=flags)
    return sc, counts, eng.download()
