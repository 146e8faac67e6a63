"""ctypes loader for the CPU oracle (oracle/surtr_oracle.cpp).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by anything under surtr_amd/.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libsurtr_oracle.so")
        if not os.path.exists(so):
            build()
        L = ctypes.CDLL(so)
        L.orc_bag_count.restype = ctypes.c_int
        L.orc_bag_bytes.restype = ctypes.c_uint64
        L.orc_bag_bytes.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.orc_bag_copy.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        L.orc_bag_count.argtypes = [ctypes.c_void_p]
        L.orc_bag_free.argtypes = [ctypes.c_void_p]
        for name in ("orc_clip", "orc_extract_faces", "orc_render", "orc_moments", "orc_islands", "orc_unit_box",
                     "orc_neighbours_from_mesh", "orc_hull_normals", "orc_kdop_planes", "orc_refit",
                     "orc_voronoi_cells", "orc_place_cells", "orc_seeds", "orc_event", "orc_regroup", "orc_convex_out_of_sphere"):
            getattr(L, name).restype = ctypes.c_void_p
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def _unbag(bag, dtypes):
    L = lib()
    out = []
    try:
        n = L.orc_bag_count(bag)
        assert n == len(dtypes), (n, len(dtypes))
        for i, dt in enumerate(dtypes):
            nb = L.orc_bag_bytes(bag, i)
            a = np.empty(nb // np.dtype(dt).itemsize, dtype=dt)
            if nb:
                L.orc_bag_copy(bag, i, _p(a))
            out.append(a)
    finally:
        L.orc_bag_free(bag)
    return out


def _solid_args(s):
    pos = np.ascontiguousarray(s["pos"], np.float32).reshape(-1, 3)
    off = np.ascontiguousarray(s["off"], np.uint32)
    nbr = np.ascontiguousarray(s["nbr"], np.int32)
    assert off.shape[0] == pos.shape[0] + 1
    return pos, off, nbr


def _solid(pos, off, nbr):
    return {"pos": pos.reshape(-1, 3), "off": off, "nbr": nbr}


def clip(solid, planes):
    pos, off, nbr = _solid_args(solid)
    pl = np.ascontiguousarray(planes, np.float32).reshape(-1, 4)
    bag = lib().orc_clip(ctypes.c_int(pos.shape[0]), _p(pos), _p(off), _p(nbr), ctypes.c_int(pl.shape[0]), _p(pl))
    return _solid(*_unbag(bag, [np.float32, np.uint32, np.int32]))


def extract_faces(solid):
    pos, off, nbr = _solid_args(solid)
    bag = lib().orc_extract_faces(ctypes.c_int(pos.shape[0]), _p(pos), _p(off), _p(nbr))
    fo, fi = _unbag(bag, [np.uint32, np.int32])
    return fo, fi


def render(solid, convex=False, colour=(0.25, 0.25, 0.25)):
    pos, off, nbr = _solid_args(solid)
    col = np.asarray(colour, np.float32)
    bag = lib().orc_render(ctypes.c_int(pos.shape[0]), _p(pos), _p(off), _p(nbr), ctypes.c_int(int(convex)), _p(col))
    vnc, idx = _unbag(bag, [np.float32, np.uint32])
    return vnc.reshape(-1, 9), idx


def moments(solid):
    pos, off, nbr = _solid_args(solid)
    bag = lib().orc_moments(ctypes.c_int(pos.shape[0]), _p(pos), _p(off), _p(nbr))
    (r,) = _unbag(bag, [np.float64])
    return float(r[0]), r[1:4].copy()


def islands(solid):
    pos, off, nbr = _solid_args(solid)
    bag = lib().orc_islands(ctypes.c_int(pos.shape[0]), _p(pos), _p(off), _p(nbr))
    lab, n = _unbag(bag, [np.int32, np.int32])
    return lab, int(n[0])


def unit_box():
    return _solid(*_unbag(lib().orc_unit_box(), [np.float32, np.uint32, np.int32]))


def neighbours_from_mesh(pos, tris):
    pos = np.ascontiguousarray(pos, np.float32).reshape(-1, 3)
    tris = np.ascontiguousarray(tris, np.int32).reshape(-1, 3)
    bag = lib().orc_neighbours_from_mesh(ctypes.c_int(pos.shape[0]), _p(pos), ctypes.c_int(tris.shape[0]), _p(tris))
    ok, off, nbr = _unbag(bag, [np.int32, np.uint32, np.int32])
    if not ok[0]:
        raise ValueError("neighbour links are not symmetric (Src/Poly.cpp:253-260)")
    return _solid(pos, off, nbr)



def transform(solid, world):
    """Poly::Transform (Src/Poly.cpp:580-585): every position through XMVector3TransformCoord(v, XMMatrixTranspose(M)) as
    ExecuteFractureRoutine applies it before an event (Src/Surtr.cpp:1846-1851).  DirectXMath is not in the reference tree; its
    published algorithm restated in float32: per row of `world` (4 x 4, row-major, translation in the last column)
    x*m0 + (y*m1 + (z*m2 + m3)), multiply-then-add without contraction, then the divide by w.  numpy restatement (no C needed:
    three multiply-adds per component)."""
    p = np.asarray(solid["pos"], np.float32).reshape(-1, 3)
    w = np.asarray(world, np.float32).reshape(4, 4)
    r = []
    for c in range(4):
        t = (p[:, 2] * w[c, 2] + w[c, 3]).astype(np.float32)
        t = (p[:, 1] * w[c, 1] + t).astype(np.float32)
        r.append((p[:, 0] * w[c, 0] + t).astype(np.float32))
    out = np.stack([(r[0] / r[3]).astype(np.float32), (r[1] / r[3]).astype(np.float32), (r[2] / r[3]).astype(np.float32)], 1)
    return dict(solid, pos=out)

def hull_normals(points, limit):
    pts = np.ascontiguousarray(points, np.float32).reshape(-1, 3)
    bag = lib().orc_hull_normals(ctypes.c_int(pts.shape[0]), _p(pts), ctypes.c_int(limit))
    (n,) = _unbag(bag, [np.float32])
    return n.reshape(-1, 3)


def kdop_planes(points, normals, ach=False, max_axis_scale=0.0, gap_inv=1.0):
    pts = np.ascontiguousarray(points, np.float32).reshape(-1, 3)
    nr = np.ascontiguousarray(normals, np.float32).reshape(-1, 3)
    bag = lib().orc_kdop_planes(ctypes.c_int(pts.shape[0]), _p(pts), ctypes.c_int(nr.shape[0]), _p(nr),
                                ctypes.c_int(int(ach)), ctypes.c_double(max_axis_scale), ctypes.c_float(gap_inv))
    (p,) = _unbag(bag, [np.float32])
    return p.reshape(-1, 4)


def links_off_the_array(reset=True):
    """How often, since the last reset, ClipPolyhedron's compaction left a link that names no vertex (Src/Poly.cpp:484-493
    through an ID of -1 or beyond the new size): past that point the reference reads outside its vertex array."""
    f = lib().orc_links_off_the_array
    f.restype = ctypes.c_long
    return int(f(ctypes.c_int(int(reset))))


def refit(convex, mesh, point_limit=4):
    cp, co, cn = _solid_args(convex)
    mp, mo, mn = _solid_args(mesh)
    bag = lib().orc_refit(ctypes.c_int(cp.shape[0]), _p(cp), _p(co), _p(cn),
                          ctypes.c_int(mp.shape[0]), _p(mp), _p(mo), _p(mn), ctypes.c_int(point_limit))
    return _solid(*_unbag(bag, [np.float32, np.uint32, np.int32]))


def voronoi_cells(seeds):
    s = np.ascontiguousarray(seeds, np.float64).reshape(-1, 3)
    bag = lib().orc_voronoi_cells(ctypes.c_int(s.shape[0]), _p(s))
    cfo, gen, fvo, verts = _unbag(bag, [np.uint32, np.int32, np.uint32, np.float64])
    return {"cell_face_off": cfo, "face_gen": gen, "face_vert_off": fvo, "verts": verts.reshape(-1, 3)}


def place_cells(v012, scale, shift):
    v = np.ascontiguousarray(v012, np.float32).reshape(-1, 9)
    sc = np.asarray(scale, np.float32)
    sh = np.asarray(shift, np.float32)
    bag = lib().orc_place_cells(ctypes.c_int(v.shape[0]), _p(v), _p(sc), _p(sh))
    (p,) = _unbag(bag, [np.float32])
    return p.reshape(-1, 4)


def seeds(n, seed=46354, mode=0, mean=1.0):
    bag = lib().orc_seeds(ctypes.c_int(n), ctypes.c_uint(seed), ctypes.c_int(mode), ctypes.c_double(mean))
    (s,) = _unbag(bag, [np.float64])
    return s.reshape(-1, 3)


def _pack_pieces(solids):
    vo = [0]
    pos, off, nbr = [], [0], []
    for s in solids:
        p, o, n = _solid_args(s)
        pos.append(p)
        base = off[-1]
        off.extend((o[1:].astype(np.int64) + base).tolist())
        nbr.append(n)
        vo.append(vo[-1] + p.shape[0])
    return (np.asarray(vo, np.uint32), np.concatenate(pos).astype(np.float32) if pos else np.zeros((0, 3), np.float32),
            np.asarray(off, np.uint32), np.concatenate(nbr).astype(np.int32) if nbr else np.zeros(0, np.int32))


def event(meshes, convexes, plane_off, planes, outside=None, refit=True, render=True, threads=1,
          cell_begin=0, cell_end=-1):
    """Runs the whole fracture event on the CPU. Returns a dict of packed fragment arrays + seconds."""
    mvo, mpos, moff, mnbr = _pack_pieces(meshes)
    cvo, cpos, coff, cnbr = _pack_pieces(convexes)
    po = np.ascontiguousarray(plane_off, np.uint32)
    pl = np.ascontiguousarray(planes, np.float32).reshape(-1, 4)
    om = None if outside is None else np.ascontiguousarray(outside, np.uint8)
    flags = (1 if refit else 0) | (2 if render else 0)
    bag = lib().orc_event(ctypes.c_int(len(meshes)), _p(mvo), _p(mpos), _p(moff), _p(mnbr),
                          _p(cvo), _p(cpos), _p(coff), _p(cnbr),
                          ctypes.c_int(po.shape[0] - 1), _p(po), _p(pl), _p(om),
                          ctypes.c_int(flags), ctypes.c_int(threads), ctypes.c_int(cell_begin), ctypes.c_int(cell_end))
    names = ["frag_ids", "mesh_vert_off", "mesh_pos", "mesh_nbr_off", "mesh_nbr",
             "conv_vert_off", "conv_pos", "conv_nbr_off", "conv_nbr", "vnc", "idx_off", "idx", "seconds"]
    dts = [np.int32, np.uint32, np.float32, np.uint32, np.int32, np.uint32, np.float32, np.uint32, np.int32,
           np.float32, np.uint32, np.uint32, np.float64]
    arrs = _unbag(bag, dts)
    out = dict(zip(names, arrs))
    out["frag_ids"] = out["frag_ids"].reshape(-1, 3)
    out["mesh_pos"] = out["mesh_pos"].reshape(-1, 3)
    out["conv_pos"] = out["conv_pos"].reshape(-1, 3)
    out["vnc"] = out["vnc"].reshape(-1, 9)
    out["seconds"] = float(out["seconds"][0])
    return out


def convex_out_of_sphere(solid, sphere_points, origin, radius):
    pos, off, nbr = _solid_args(solid)
    sp = np.ascontiguousarray(sphere_points, np.float32).reshape(-1, 3)
    org = np.asarray(origin, np.float32)
    bag = lib().orc_convex_out_of_sphere(ctypes.c_int(pos.shape[0]), _p(pos), _p(off), _p(nbr), ctypes.c_int(sp.shape[0]), _p(sp),
                                         _p(org), ctypes.c_float(radius))
    (f,) = _unbag(bag, [np.int32])
    return bool(f[0])


def regroup(convexes, piece_cell, n_outside=0, partial=False, sphere_points=None, origin=(0, 0, 0), radius=1.0):
    cvo, cpos, coff, cnbr = _pack_pieces(convexes)
    pc = np.ascontiguousarray(piece_cell, np.int32)
    sp = np.zeros((0, 3), np.float32) if sphere_points is None else np.ascontiguousarray(sphere_points, np.float32).reshape(-1, 3)
    org = np.asarray(origin, np.float32)
    bag = lib().orc_regroup(ctypes.c_int(len(convexes)), ctypes.c_int(n_outside), _p(pc), _p(cvo), _p(cpos), _p(coff), _p(cnbr),
                            ctypes.c_int(int(partial)), ctypes.c_int(sp.shape[0]), _p(sp), _p(org), ctypes.c_float(radius))
    co, cp = _unbag(bag, [np.uint32, np.int32])
    return co, cp
