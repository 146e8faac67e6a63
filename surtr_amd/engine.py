"""ctypes binding of libsurtr_hip.so (include/surtr_hip.h) for tests and bench.

This is plumbing only: numpy arrays in, numpy arrays out, every call goes
through the C ABI.  There is no CPU fallback -- if the shared library or a HIP
device is missing the calls raise.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

OK, E_INVALID, E_TOPOLOGY, E_CAPACITY, E_HIP, E_STATE, E_NOGPU = range(7)
EVT_REFIT, EVT_RENDER = 1, 2


class SurtrError(RuntimeError):
    def __init__(self, code, msg=""):
        self.code = code
        RuntimeError.__init__(self, "surtr error %d: %s %s" % (code, _strerror(code), msg))


class Counts(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint32) for n in
                ("n_frag", "mesh_verts", "mesh_nbrs", "conv_verts", "conv_nbrs", "n_idx", "n_pairs", "status", "n_failed")]


class Fragments(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in
                ("frag_ids", "mesh_vert_off", "mesh_pos", "mesh_nbr_off", "mesh_nbr", "conv_vert_off", "conv_pos",
                 "conv_nbr_off", "conv_nbr", "vnc", "idx_off", "idx", "frag_status")]


def lib_path():
    return os.path.join(_HERE, "libsurtr_hip.so")


def _configure(L):
    L.surtr_strerror.restype = ctypes.c_char_p
    L.surtr_last_error.restype = ctypes.c_char_p
    L.surtr_last_error.argtypes = [ctypes.c_void_p]
    L.surtr_event_blob_bytes.restype = ctypes.c_size_t
    L.surtr_destroy.argtypes = [ctypes.c_void_p]
    L.surtr_destroy.restype = None
    return L


def _use_library_for_tests(path):
    """tests/ only: bind the single-lane CPU emulation of the kernels (tests/emul/libsurtr_emul.so)
    so kernel logic can be checked without a GPU.  Never called by product code."""
    global _LIB
    _LIB = _configure(ctypes.CDLL(path)) if path else None


def _share_hip_runtime_with_torch():
    """PyTorch wheels bundle their own libamdhip64 / libhsa-runtime64.  If libsurtr_hip.so is loaded first it brings in
    /opt/rocm's runtime, a later `import torch` brings in the bundled one, and the second ROCr runtime of a process cannot open
    the GPU ("No HIP GPUs are available").  A process that may use both (tests, bench.py, multigpu.py hand torch buffers to the
    engine) therefore loads the runtime torch would load, first: libsurtr_hip.so's libamdhip64.so.N then resolves to it by
    soname, which is what happens anyway when torch is imported before the engine.  Without torch installed this does nothing."""
    import sys, importlib.util
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            pass


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(lib_path()):
            raise ImportError("libsurtr_hip.so is not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
        _share_hip_runtime_with_torch()
        L = _configure(ctypes.CDLL(lib_path()))
        _LIB = L
    return _LIB


def _strerror(code):
    try:
        return lib().surtr_strerror(ctypes.c_int(code)).decode()
    except Exception:
        return "?"


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _solid_arrays(s):
    return (np.ascontiguousarray(s["pos"], np.float32).reshape(-1, 3), np.ascontiguousarray(s["off"], np.uint32),
            np.ascontiguousarray(s["nbr"], np.int32))


def pack_solids(solids):
    """list of {'pos','off','nbr'} -> (vert_off, pos, nbr_off (global), nbr)"""
    vo = [0]
    pos, off, nbr = [], [np.zeros(1, np.int64)], []
    base = 0
    for s in solids:
        p = np.ascontiguousarray(s["pos"], np.float32).reshape(-1, 3)
        o = np.asarray(s["off"], np.int64)
        pos.append(p)
        off.append(o[1:] + base)
        base += int(o[-1])
        nbr.append(np.asarray(s["nbr"], np.int32))
        vo.append(vo[-1] + p.shape[0])
    return (np.asarray(vo, np.uint32), np.ascontiguousarray(np.concatenate(pos), np.float32),
            np.concatenate(off).astype(np.uint32), np.ascontiguousarray(np.concatenate(nbr), np.int32))


class Engine:
    """One context per GPU (surtr_create / surtr_destroy)."""

    def __init__(self, device=0, stream=None):
        self._h = ctypes.c_void_p()
        rc = lib().surtr_create(ctypes.c_int(device), ctypes.byref(self._h))
        if rc:
            raise SurtrError(rc)
        if stream is not None:
            self.set_stream(stream)

    def close(self):
        if self._h:
            lib().surtr_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc:
            raise SurtrError(rc, lib().surtr_last_error(self._h).decode())

    def set_stream(self, stream_ptr):
        self._ck(lib().surtr_set_stream(self._h, ctypes.c_void_p(stream_ptr)))

    def set_scratch(self, max_verts, max_nbrs):
        self._ck(lib().surtr_set_scratch(self._h, ctypes.c_uint32(max_verts), ctypes.c_uint32(max_nbrs)))

    def set_arena(self, verts, nbrs, idx):
        self._ck(lib().surtr_set_arena(self._h, ctypes.c_uint64(verts), ctypes.c_uint64(nbrs), ctypes.c_uint64(idx)))

    def set_profiling(self, on=True):
        self._ck(lib().surtr_set_profiling(self._h, ctypes.c_int(int(on))))

    def kernel_times(self):
        ms = (ctypes.c_float * 16)()
        self._ck(lib().surtr_kernel_times(self._h, ms))
        names = ("clip_pairs", "frag_table", "refit", "faces", "out_scan", "pack", "clip_convex", "prep_pairs", "clip_pairs_big", "clip_pairs_half", "clip_pairs_retry", "clip_pairs_wave", "clip_pairs_rec", "clip_pairs_catch")
        return {n: float(ms[i]) for i, n in enumerate(names)}

    def set_events_in_flight(self, n):
        """Tells the context how many contexts the host keeps busy on this GPU (include/surtr_hip.h: surtr_set_events_in_flight)."""
        self._ck(lib().surtr_set_events_in_flight(self._h, ctypes.c_uint32(int(n))))

    def kernel_history(self):
        """Durations (ms) of the Mesh clip kernel over the last events, oldest first (surtr_kernel_history)."""
        ms = (ctypes.c_float * 16)(); slot = (ctypes.c_int * 16)(); n = ctypes.c_uint32(0)
        self._ck(lib().surtr_kernel_history(self._h, ms, slot, ctypes.byref(n)))
        return [float(ms[i]) for i in range(n.value)]

    def pair_status(self, n_pairs):
        """Status per pair of the last event (include/surtr_hip.h: surtr_pair_status)."""
        out = np.zeros(n_pairs, dtype=np.uint32)
        self._ck(lib().surtr_pair_status(self._h, ctypes.c_uint32(n_pairs), _p(out)))
        return out

    def pair_costs(self, n_pairs):
        """Cost estimate per pair of the last event (include/surtr_hip.h: surtr_event_pair_costs)."""
        out = np.zeros(n_pairs, dtype=np.uint32)
        self._ck(lib().surtr_event_pair_costs(self._h, ctypes.c_uint32(n_pairs), _p(out)))
        return out

    def queue_stats(self):
        """Device-side counters of the last event (include/surtr_hip.h: surtr_queue_stats)."""
        out = (ctypes.c_uint32 * 128)()
        self._ck(lib().surtr_queue_stats(self._h, out))
        return np.frombuffer(out, dtype=np.uint32).copy()

    def upload_pieces(self, meshes, convexes):
        assert len(meshes) == len(convexes)
        m = pack_solids(meshes)
        c = pack_solids(convexes)
        self._ck(lib().surtr_upload_pieces(self._h, ctypes.c_uint32(len(meshes)), _p(m[0]), _p(m[1]), _p(m[2]), _p(m[3]),
                                           _p(c[0]), _p(c[1]), _p(c[2]), _p(c[3])))

    def build_cells(self, seeds, group_seed_off=None):
        """surtr_build_cells: Voronoi cells of the seeds on the device, installed as the pattern.  Returns (n_faces, n_face_verts)."""
        s = np.ascontiguousarray(seeds, np.float64).reshape(-1, 3)
        go = np.array([0, s.shape[0]], np.uint32) if group_seed_off is None else np.ascontiguousarray(group_seed_off, np.uint32)
        nf, nfv = ctypes.c_uint32(), ctypes.c_uint32()
        self._ck(lib().surtr_build_cells(self._h, ctypes.c_uint32(go.shape[0] - 1), _p(go), _p(s), ctypes.byref(nf), ctypes.byref(nfv)))
        self._cells_n = (s.shape[0], nf.value, nfv.value)
        return nf.value, nfv.value

    def download_cells(self):
        """The cells of the last build_cells in the layout of engine.voronoi_cells (+ 'v012')."""
        n, nf, nfv = self._cells_n
        cfo = np.zeros(n + 1, np.uint32); gen = np.zeros(nf, np.int32); fvo = np.zeros(nf + 1, np.uint32)
        verts = np.zeros((nfv, 3), np.float64); v012 = np.zeros((nf, 9), np.float32)
        self._ck(lib().surtr_download_cells(self._h, _p(cfo), _p(gen), _p(fvo), _p(verts), _p(v012)))
        return {"cell_face_off": cfo, "face_gen": gen, "face_vert_off": fvo, "verts": verts, "v012": v012}

    def neighbors_from_mesh(self, pos, tris):
        """Poly::ExtractNeighborFromMesh on the device (surtr_neighbors_from_mesh_dev) -> (solid, kernel milliseconds)."""
        pos = np.ascontiguousarray(pos, np.float32).reshape(-1, 3)
        tris = np.ascontiguousarray(tris, np.int32).reshape(-1, 3)
        off = np.zeros(pos.shape[0] + 1, np.uint32)
        nbr = np.zeros(3 * tris.shape[0] + 4, np.int32)
        ms = ctypes.c_float()
        self._ck(lib().surtr_neighbors_from_mesh_dev(self._h, ctypes.c_uint32(pos.shape[0]), ctypes.c_uint32(tris.shape[0]), _p(tris), _p(off), _p(nbr),
                                                     ctypes.byref(ms)))
        return {"pos": pos, "off": off, "nbr": nbr[:off[-1]].copy()}, float(ms.value)

    def upload_pattern(self, face_off, v012):
        fo = np.ascontiguousarray(face_off, np.uint32)
        v = np.ascontiguousarray(v012, np.float32).reshape(-1, 9)
        assert v.shape[0] == fo[-1]
        self._ck(lib().surtr_upload_pattern(self._h, ctypes.c_uint32(fo.shape[0] - 1), _p(fo), _p(v)))

    def place_cells(self, scale, translate):
        s = np.ascontiguousarray(scale, np.float32)
        t = np.ascontiguousarray(translate, np.float32)
        self._ck(lib().surtr_place_cells(self._h, _p(s), _p(t)))

    def upload_planes(self, plane_off, planes):
        po = np.ascontiguousarray(plane_off, np.uint32)
        pl = np.ascontiguousarray(planes, np.float32).reshape(-1, 4)
        self._ck(lib().surtr_upload_planes(self._h, ctypes.c_uint32(po.shape[0] - 1), _p(po), _p(pl)))

    def fracture_event(self, cell_begin, cell_end, outside=None, flags=EVT_REFIT | EVT_RENDER):
        c = Counts()
        om = None if outside is None else np.ascontiguousarray(outside, np.uint8)
        self._ck(lib().surtr_fracture_event(self._h, ctypes.c_uint32(cell_begin), ctypes.c_uint32(cell_end), _p(om),
                                            ctypes.c_uint32(flags), ctypes.byref(c)))
        return c

    def fracture_event_async(self, cell_begin, cell_end, outside=None, flags=EVT_REFIT | EVT_RENDER):
        om = None if outside is None else np.ascontiguousarray(outside, np.uint8)
        self._ck(lib().surtr_fracture_event_async(self._h, ctypes.c_uint32(cell_begin), ctypes.c_uint32(cell_end), _p(om),
                                                  ctypes.c_uint32(flags)))

    def place_cells_groups(self, group_cell_off, scales, translates):
        go = np.ascontiguousarray(group_cell_off, np.uint32)
        s = np.ascontiguousarray(scales, np.float32).reshape(-1, 3)
        t = np.ascontiguousarray(translates, np.float32).reshape(-1, 3)
        assert s.shape[0] == go.shape[0] - 1 == t.shape[0]
        self._ck(lib().surtr_place_cells_groups(self._h, ctypes.c_uint32(s.shape[0]), _p(go), _p(s), _p(t)))

    def place_cells_in_pieces(self, group_cell_off):
        go = np.ascontiguousarray(group_cell_off, np.uint32)
        self._ck(lib().surtr_place_cells_in_pieces(self._h, ctypes.c_uint32(go.shape[0] - 1), _p(go)))

    def fracture_pairs(self, pair_cell, pair_piece, flags=EVT_REFIT | EVT_RENDER):
        pc = np.ascontiguousarray(pair_cell, np.uint32)
        pp = np.ascontiguousarray(pair_piece, np.uint32)
        assert pc.shape == pp.shape
        self._ck(lib().surtr_fracture_pairs_async(self._h, ctypes.c_uint32(pc.shape[0]), _p(pc), _p(pp), ctypes.c_uint32(flags)))
        return self.event_counts()

    def event_regroup(self, partial=False, sphere_points=None, origin=(0, 0, 0), radius=1.0):
        """surtr_event_regroup: bind sets + MergeOutOfImpact + HandleConvexIsland of the last event, on the device."""
        sp = np.zeros((0, 3), np.float32) if sphere_points is None else np.ascontiguousarray(sphere_points, np.float32).reshape(-1, 3)
        org = np.ascontiguousarray(origin, np.float32)
        n, nc = ctypes.c_uint32(), ctypes.c_uint32()
        args = [self._h, ctypes.c_int(int(partial)), ctypes.c_uint32(sp.shape[0]), _p(sp), _p(org), ctypes.c_float(radius)]
        self._ck(lib().surtr_event_regroup(*args, ctypes.byref(n), ctypes.byref(nc), None, None))
        co = np.zeros(n.value + 2, np.uint32); cp = np.zeros(max(n.value, 1), np.int32)
        self._ck(lib().surtr_event_regroup(*args, ctypes.byref(n), ctypes.byref(nc), _p(co), _p(cp)))
        return co[:nc.value + 1].copy(), cp[:co[nc.value]].copy()

    def event_refit(self):
        self._ck(lib().surtr_event_refit(self._h))

    def event_counts(self):
        c = Counts()
        self._ck(lib().surtr_event_counts(self._h, ctypes.byref(c)))
        return c

    def pack_dev(self, dev_ptr, capacity):
        self._ck(lib().surtr_event_pack_dev(self._h, ctypes.c_void_p(dev_ptr), ctypes.c_size_t(capacity)))

    def download(self):
        c = self.event_counts()
        out = alloc_fragments(c)
        fr = _frag_struct(out)
        self._ck(lib().surtr_event_download(self._h, ctypes.byref(fr)))
        return shape_fragments(out)

    def load_fragments(self, meshes, convexes, frag_ids=None):
        """surtr_load_fragments: host pieces presented as the fragments of an event."""
        m = pack_solids(meshes)
        c = pack_solids(convexes)
        ids = None if frag_ids is None else np.ascontiguousarray(frag_ids, np.int32)
        self._ck(lib().surtr_load_fragments(self._h, ctypes.c_uint32(len(meshes)), _p(m[0]), _p(m[1]), _p(m[2]), _p(m[3]),
                                            _p(c[0]), _p(c[1]), _p(c[2]), _p(c[3]), _p(ids)))

    def event_triangulate(self, is_convex=False):
        self._ck(lib().surtr_event_triangulate(self._h, ctypes.c_int(int(is_convex))))

    def refit_solid(self, mesh, convex):
        """m_refittingTask for one Piece (surtr_refit_solid) -> the refitted Convex."""
        mp, mo, mn = _solid_arrays(mesh)
        cp, co, cn = _solid_arrays(convex)
        nv, nh = ctypes.c_uint32(), ctypes.c_uint32()
        args = [self._h, ctypes.c_uint32(mp.shape[0]), _p(mp), _p(mo), _p(mn), ctypes.c_uint32(cp.shape[0]), _p(cp), _p(co), _p(cn)]
        self._ck(lib().surtr_refit_solid(*args, ctypes.byref(nv), ctypes.byref(nh), None, None, None))
        opos = np.zeros((nv.value, 3), np.float32); ooff = np.zeros(nv.value + 1, np.uint32); onbr = np.zeros(nh.value, np.int32)
        self._ck(lib().surtr_refit_solid(*args, ctypes.byref(nv), ctypes.byref(nh), _p(opos), _p(ooff), _p(onbr)))
        return {"pos": opos, "off": ooff, "nbr": onbr}

    def extract_faces(self, solid):
        """Poly::ExtractFaces (surtr_extract_faces) -> (face_off, face_idx)."""
        p, o, n = _solid_arrays(solid)
        nf, ni = ctypes.c_uint32(), ctypes.c_uint32()
        args = [self._h, ctypes.c_uint32(p.shape[0]), _p(p), _p(o), _p(n)]
        self._ck(lib().surtr_extract_faces(*args, ctypes.byref(nf), ctypes.byref(ni), None, None))
        fo = np.zeros(nf.value + 1, np.uint32); fi = np.zeros(ni.value, np.int32)
        self._ck(lib().surtr_extract_faces(*args, ctypes.byref(nf), ctypes.byref(ni), _p(fo), _p(fi)))
        return fo, fi

    def triangulate(self, solid, is_convex=False, color=None):
        """Poly::ExtractFaces + Poly::RenderPolyhedron (surtr_triangulate) -> (vnc f32[n,9], idx u32[...])."""
        p, o, n = _solid_arrays(solid)
        col = None if color is None else np.ascontiguousarray(color, np.float32)
        ni = ctypes.c_uint32()
        args = [self._h, ctypes.c_uint32(p.shape[0]), _p(p), _p(o), _p(n), ctypes.c_int(int(is_convex)), _p(col)]
        self._ck(lib().surtr_triangulate(*args, None, ctypes.byref(ni), None))
        vnc = np.zeros((p.shape[0], 9), np.float32); idx = np.zeros(ni.value, np.uint32)
        self._ck(lib().surtr_triangulate(*args, _p(vnc), ctypes.byref(ni), _p(idx)))
        return vnc, idx

    def transform_pieces(self, world):
        """Poly::Transform of every resident piece (surtr_transform_pieces); world: f32[n,4,4] row-major XMMATRIX."""
        w = np.ascontiguousarray(world, np.float32).reshape(-1, 16)
        self._ck(lib().surtr_transform_pieces(self._h, ctypes.c_uint32(w.shape[0]), _p(w)))

    def pieces_from_event(self, keep=None):
        """surtr_pieces_from_event: the last event's fragments become the resident pieces; returns their number."""
        k = None if keep is None else np.ascontiguousarray(keep, np.uint8)
        n = ctypes.c_uint32()
        self._ck(lib().surtr_pieces_from_event(self._h, _p(k), ctypes.byref(n)))
        return n.value

    def upload_stats(self):
        ms, na = ctypes.c_float(), ctypes.c_uint32()
        self._ck(lib().surtr_upload_stats(self._h, ctypes.byref(ms), ctypes.byref(na)))
        return float(ms.value), int(na.value)

    def clip_polyhedron(self, solid, planes):
        pos = np.ascontiguousarray(solid["pos"], np.float32).reshape(-1, 3)
        off = np.ascontiguousarray(solid["off"], np.uint32)
        nbr = np.ascontiguousarray(solid["nbr"], np.int32)
        pl = np.ascontiguousarray(planes, np.float32).reshape(-1, 4)
        nv = ctypes.c_uint32()
        nh = ctypes.c_uint32()
        args = [self._h, ctypes.c_uint32(pos.shape[0]), _p(pos), _p(off), _p(nbr), ctypes.c_uint32(pl.shape[0]), _p(pl)]
        self._ck(lib().surtr_clip_polyhedron(*args, ctypes.byref(nv), ctypes.byref(nh), None, None, None))
        opos = np.zeros((nv.value, 3), np.float32)
        ooff = np.zeros(nv.value + 1, np.uint32)
        onbr = np.zeros(nh.value, np.int32)
        self._ck(lib().surtr_clip_polyhedron(*args, ctypes.byref(nv), ctypes.byref(nh), _p(opos), _p(ooff), _p(onbr)))
        return {"pos": opos, "off": ooff, "nbr": onbr}


_FIELDS = [("frag_ids", np.int32, lambda c: 3 * c.n_frag), ("mesh_vert_off", np.uint32, lambda c: c.n_frag + 1),
           ("mesh_pos", np.float32, lambda c: 3 * c.mesh_verts), ("mesh_nbr_off", np.uint32, lambda c: c.mesh_verts + 1),
           ("mesh_nbr", np.int32, lambda c: c.mesh_nbrs), ("conv_vert_off", np.uint32, lambda c: c.n_frag + 1),
           ("conv_pos", np.float32, lambda c: 3 * c.conv_verts), ("conv_nbr_off", np.uint32, lambda c: c.conv_verts + 1),
           ("conv_nbr", np.int32, lambda c: c.conv_nbrs), ("vnc", np.float32, lambda c: 9 * c.mesh_verts),
           ("idx_off", np.uint32, lambda c: c.n_frag + 1), ("idx", np.uint32, lambda c: c.n_idx),
           ("frag_status", np.uint32, lambda c: c.n_frag)]


def alloc_fragments(c):
    return {name: np.zeros(max(int(size(c)), 0), dt) for name, dt, size in _FIELDS}


def _frag_struct(out):
    fr = Fragments()
    for name, _, _ in _FIELDS:
        setattr(fr, name, out[name].ctypes.data)
    return fr


def shape_fragments(out):
    out = dict(out)
    out["frag_ids"] = out["frag_ids"].reshape(-1, 3)
    out["mesh_pos"] = out["mesh_pos"].reshape(-1, 3)
    out["conv_pos"] = out["conv_pos"].reshape(-1, 3)
    out["vnc"] = out["vnc"].reshape(-1, 9)
    return out


def unpack_blob(host_blob):
    """host_blob: contiguous uint8 numpy array holding one packed fragment blob."""
    b = np.ascontiguousarray(host_blob, np.uint8)
    c = Counts()
    rc = lib().surtr_blob_unpack_host(_p(b), ctypes.c_size_t(b.nbytes), ctypes.byref(c), None)
    if rc:
        raise SurtrError(rc)
    out = alloc_fragments(c)
    fr = _frag_struct(out)
    rc = lib().surtr_blob_unpack_host(_p(b), ctypes.c_size_t(b.nbytes), ctypes.byref(c), ctypes.byref(fr))
    if rc:
        raise SurtrError(rc)
    return c, shape_fragments(out)


def blob_bytes(counts):
    return int(lib().surtr_event_blob_bytes(ctypes.byref(counts)))


def neighbors_from_mesh(pos, tris):
    """Poly::ExtractNeighborFromMesh through the C ABI (host helper)."""
    pos = np.ascontiguousarray(pos, np.float32).reshape(-1, 3)
    tris = np.ascontiguousarray(tris, np.int32).reshape(-1, 3)
    off = np.zeros(pos.shape[0] + 1, np.uint32)
    nbr = np.zeros(6 * tris.shape[0], np.int32)
    rc = lib().surtr_neighbors_from_mesh(ctypes.c_uint32(pos.shape[0]), ctypes.c_uint32(tris.shape[0]), _p(tris), _p(off), _p(nbr))
    if rc:
        raise SurtrError(rc)
    return {"pos": pos, "off": off, "nbr": nbr[:off[-1]].copy()}


def voronoi_cells(seeds):
    s = np.ascontiguousarray(seeds, np.float64).reshape(-1, 3)
    nf = ctypes.c_uint32()
    nfv = ctypes.c_uint32()
    rc = lib().surtr_voronoi_cells(ctypes.c_uint32(s.shape[0]), _p(s), ctypes.byref(nf), ctypes.byref(nfv), None, None, None, None)
    if rc:
        raise SurtrError(rc)
    cfo = np.zeros(s.shape[0] + 1, np.uint32)
    gen = np.zeros(nf.value, np.int32)
    fvo = np.zeros(nf.value + 1, np.uint32)
    verts = np.zeros((nfv.value, 3), np.float64)
    rc = lib().surtr_voronoi_cells(ctypes.c_uint32(s.shape[0]), _p(s), ctypes.byref(nf), ctypes.byref(nfv), _p(cfo), _p(gen), _p(fvo),
                                   _p(verts))
    if rc:
        raise SurtrError(rc)
    return {"cell_face_off": cfo, "face_gen": gen, "face_vert_off": fvo, "verts": verts}


def pattern_from_cells(cells):
    """First three vertices of every face, narrowed to float (what ConstructFacePlane reads)."""
    fvo = cells["face_vert_off"]
    v = cells["verts"].astype(np.float32)
    nf = fvo.shape[0] - 1
    v012 = np.stack([v[fvo[:-1] + k] for k in range(3)], 1).reshape(nf, 9)
    return cells["cell_face_off"], np.ascontiguousarray(v012)


def merge_fragments(parts):
    """Concatenates the fragment sets of consecutive cell blocks (one per rank, in rank order) into one
    set with rebased offsets: the cell-major order of Surtr::ApplyFracture (Src/Surtr.cpp:2133-2146)."""
    out = {}
    out["frag_ids"] = np.concatenate([p["frag_ids"].reshape(-1, 3) for p in parts])
    for pre in ("mesh", "conv"):
        vo, no, vbase, nbase = [np.zeros(1, np.uint32)], [np.zeros(1, np.uint32)], 0, 0
        for p in parts:
            pvo, pno = p[pre + "_vert_off"].astype(np.int64), p[pre + "_nbr_off"].astype(np.int64)
            vo.append((pvo[1:] + vbase).astype(np.uint32))
            no.append((pno[1:] + nbase).astype(np.uint32))
            vbase += int(pvo[-1])
            nbase += int(pno[-1])
        out[pre + "_vert_off"] = np.concatenate(vo)
        out[pre + "_nbr_off"] = np.concatenate(no)
        out[pre + "_pos"] = np.concatenate([p[pre + "_pos"].reshape(-1, 3) for p in parts])
        out[pre + "_nbr"] = np.concatenate([p[pre + "_nbr"] for p in parts])
    io, ibase = [np.zeros(1, np.uint32)], 0
    for p in parts:
        pio = p["idx_off"].astype(np.int64)
        io.append((pio[1:] + ibase).astype(np.uint32))
        ibase += int(pio[-1])
    out["idx_off"] = np.concatenate(io)
    out["idx"] = np.concatenate([p["idx"] for p in parts])
    out["vnc"] = np.concatenate([p["vnc"].reshape(-1, 9) for p in parts])
    return out


def cell_block(rank, world, n_cells):
    """Contiguous cell block of a rank: [floor(r*C/G), floor((r+1)*C/G))."""
    return (rank * n_cells) // world, ((rank + 1) * n_cells) // world


def balanced_blocks(costs, world):
    """Cut points of `world` CONTIGUOUS blocks of units (cells of an event, or the pairs of a pair list) with costs `costs`:
    block r = [cuts[r], cuts[r + 1]).  Order is untouched -- results concatenated in rank order are still in unit order
    (Src/Surtr.cpp:2133-2146) -- only where the cuts fall changes: cut r sits where the running cost passes r / world of the total
    (SURVEY.md section 8e).  Deterministic in its inputs, so every rank computes the same cuts from the same costs."""
    c = np.asarray(costs, np.float64)
    n = c.shape[0]
    if n == 0 or world <= 1:
        return [0] + [n] * max(world, 1)
    cum = np.concatenate([[0.0], np.cumsum(c)])
    cuts = [0]
    for r in range(1, world):
        k = int(np.searchsorted(cum, cum[-1] * r / world, side="left"))
        # the unit whose cost straddles the target goes to the side that leaves the smaller excess
        if k > 0 and k <= n and abs(cum[k - 1] - cum[-1] * r / world) <= abs(cum[min(k, n)] - cum[-1] * r / world):
            k -= 1
        cuts.append(min(max(k, cuts[-1]), n))
    cuts.append(n)
    return cuts


def pair_block(rank, world, n_pairs):
    """Contiguous block of a (fragment-major) pair list for a rank: [floor(r*P/G), floor((r+1)*P/G)) -- the sharding of a
    recursive refracture (BASELINE configs[4]): results concatenated in rank order are in list
    order, i.e. the order in which the reference's per-compound fan-outs would return them (Src/Surtr.cpp:2129-2146)."""
    return (rank * n_pairs) // world, ((rank + 1) * n_pairs) // world


def hull_normals(points, limit):
    """VMACH::ConvexHull(points, limit) face normals (host helper)."""
    p = np.ascontiguousarray(points, np.float32).reshape(-1, 3)
    cnt = ctypes.c_uint32()
    rc = lib().surtr_hull_normals(ctypes.c_uint32(p.shape[0]), _p(p), ctypes.c_uint32(limit), ctypes.c_uint32(0), None, ctypes.byref(cnt))
    if rc:
        raise SurtrError(rc)
    out = np.zeros((cnt.value, 3), np.float32)
    rc = lib().surtr_hull_normals(ctypes.c_uint32(p.shape[0]), _p(p), ctypes.c_uint32(limit), ctypes.c_uint32(cnt.value), _p(out), ctypes.byref(cnt))
    if rc:
        raise SurtrError(rc)
    return out


def kdop_ach_planes(points, normals, max_axis_scale, plane_gap_inv=2000.0):
    p = np.ascontiguousarray(points, np.float32).reshape(-1, 3)
    n = np.ascontiguousarray(normals, np.float32).reshape(-1, 3)
    out = np.zeros((2 * n.shape[0], 4), np.float32)
    rc = lib().surtr_kdop_ach_planes(ctypes.c_uint32(p.shape[0]), _p(p), ctypes.c_uint32(n.shape[0]), _p(n), ctypes.c_double(max_axis_scale),
                                     ctypes.c_float(plane_gap_inv), _p(out))
    if rc:
        raise SurtrError(rc)
    return out


def convex_out_of_sphere(solid, sphere_points, origin, radius):
    pos = np.ascontiguousarray(solid["pos"], np.float32).reshape(-1, 3)
    off = np.ascontiguousarray(solid["off"], np.uint32)
    nbr = np.ascontiguousarray(solid["nbr"], np.int32)
    sp = np.ascontiguousarray(sphere_points, np.float32).reshape(-1, 3)
    org = np.ascontiguousarray(origin, np.float32)
    out = ctypes.c_int()
    rc = lib().surtr_convex_out_of_sphere(ctypes.c_uint32(pos.shape[0]), _p(pos), _p(off), _p(nbr), ctypes.c_uint32(sp.shape[0]), _p(sp),
                                          _p(org), ctypes.c_float(radius), ctypes.byref(out))
    if rc:
        raise SurtrError(rc)
    return bool(out.value)


def regroup(convexes, piece_cell, n_outside=0, partial=False, sphere_points=None, origin=(0, 0, 0), radius=1.0):
    """Bind sets + MergeOutOfImpact + HandleConvexIsland on un-refitted Convex solids -> (compound_off, compound_piece)."""
    cvo, cpos, coff, cnbr = pack_solids(convexes)
    pc = np.ascontiguousarray(piece_cell, np.int32)
    sp = np.zeros((0, 3), np.float32) if sphere_points is None else np.ascontiguousarray(sphere_points, np.float32).reshape(-1, 3)
    org = np.ascontiguousarray(origin, np.float32)
    n = len(convexes)
    co = np.zeros(n + 2, np.uint32)
    cp = np.zeros(max(n, 1), np.int32)
    nc = ctypes.c_uint32()
    rc = lib().surtr_regroup(ctypes.c_uint32(n), ctypes.c_uint32(n_outside), _p(pc), _p(cvo), _p(cpos), _p(coff), _p(cnbr),
                             ctypes.c_int(int(partial)), ctypes.c_uint32(sp.shape[0]), _p(sp), _p(org), ctypes.c_float(radius),
                             ctypes.byref(nc), _p(co), _p(cp))
    if rc:
        raise SurtrError(rc)
    return co[:nc.value + 1].copy(), cp[:co[nc.value]].copy()


def moments(solid):
    """Poly::Moments (Src/Poly.cpp:55-87) of one solid: (volume, centroid)."""
    pos = np.ascontiguousarray(solid["pos"], np.float32).reshape(-1, 3)
    off = np.ascontiguousarray(solid["off"], np.uint32)
    nbr = np.ascontiguousarray(solid["nbr"], np.int32)
    vol = ctypes.c_double(0.0)
    cen = np.zeros(3, np.float32)
    rc = lib().surtr_moments(ctypes.c_uint32(pos.shape[0]), _p(pos), _p(off), _p(nbr), ctypes.byref(vol), _p(cen))
    if rc:
        raise SurtrError(rc)
    return float(vol.value), cen


def read_obj(path, scale=(1, 1, 1), translate=(0, 0, 0)):
    """Surtr::LoadModelData conventions (Src/Surtr.cpp:2683-2727) for a Wavefront OBJ: (verts f32[n,3], tris i32[m,3])."""
    sc = np.asarray(scale, np.float32); tr = np.asarray(translate, np.float32)
    nv, nt = ctypes.c_uint32(0), ctypes.c_uint32(0)
    rc = lib().surtr_read_obj(path.encode(), _p(sc), _p(tr), ctypes.c_uint32(0), ctypes.c_uint32(0), None, None, ctypes.byref(nv), ctypes.byref(nt))
    if rc:
        raise SurtrError(rc)
    pos = np.zeros((nv.value, 3), np.float32); tris = np.zeros((nt.value, 3), np.int32)
    rc = lib().surtr_read_obj(path.encode(), _p(sc), _p(tr), nv, nt, _p(pos), _p(tris), ctypes.byref(nv), ctypes.byref(nt))
    if rc:
        raise SurtrError(rc)
    return pos, tris


def write_obj(path, fragments):
    """One OBJ object per fragment from the render buffers of engine.download()."""
    fr = fragments
    ids = np.ascontiguousarray(fr["frag_ids"], np.int32)
    rc = lib().surtr_write_obj(path.encode(), ctypes.c_uint32(ids.shape[0]), _p(ids), _p(np.ascontiguousarray(fr["mesh_vert_off"], np.uint32)),
                               _p(np.ascontiguousarray(fr["vnc"], np.float32)), _p(np.ascontiguousarray(fr["idx_off"], np.uint32)),
                               _p(np.ascontiguousarray(fr["idx"], np.uint32)))
    if rc:
        raise SurtrError(rc)
