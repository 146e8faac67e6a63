"""literal_clip.h -- the one-lane, statement-by-statement ClipPolyhedron the kernels fall back to when the parallel relink
meets a face walk it cannot follow (Src/Poly.cpp:389-394 running into its step bound on a degenerate sliver).

CPU tier: the header alone (tests/emul/literal_probe.cpp) against the oracle on ordinary solids, where both clippers must agree
bit for bit; then the fuzz-found degenerate pair (refracture fuzz seed 555, case 110 -- tests/golden/
degenerate_walk_bound_convex.npz: a piece whose Mesh is four coincident-vertex slivers and whose Convex lists neighbours
twice) through the engine entry points on the emulation.  The GPU tier repeats the second half (tests/test_gpu_parity.py).
"""
import ctypes
import os

import numpy as np
import pytest

from helpers import fragment
from surtr_amd import scenes


def assert_solid_equal(got, ref):
    """The literal path runs the reference's statements in the reference's order: bit-exact, coordinates included."""
    assert got["pos"].shape[0] == ref["pos"].shape[0]
    assert np.array_equal(got["off"], ref["off"]) and np.array_equal(got["nbr"], ref["nbr"])
    assert np.array_equal(np.asarray(got["pos"], np.float32).reshape(-1, 3), np.asarray(ref["pos"], np.float32).reshape(-1, 3))

HERE = os.path.dirname(os.path.abspath(__file__))


def _probe(emul_lib_path):
    lib = ctypes.CDLL(os.path.join(os.path.dirname(emul_lib_path), "libliteral_probe.so"))
    lib.literal_probe.restype = ctypes.c_int
    return lib


def _literal(lib, solid, planes, cap_v=512):
    pos = np.ascontiguousarray(solid["pos"], np.float32); off = np.ascontiguousarray(solid["off"], np.uint32)
    nbr = np.ascontiguousarray(solid["nbr"], np.int32); pl = np.ascontiguousarray(planes, np.float32).reshape(-1, 4)
    opos = np.zeros((cap_v, 3), np.float32); ooff = np.zeros(cap_v + 1, np.uint32); onbr = np.zeros(cap_v * 32, np.int32)
    n = ctypes.c_uint32(0)
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    rc = lib.literal_probe(ctypes.c_uint32(pos.shape[0]), P(pos), P(off), P(nbr), ctypes.c_uint32(pl.shape[0]), P(pl),
                           ctypes.c_uint32(cap_v), P(opos), P(ooff), P(onbr), ctypes.byref(n))
    if rc != 0:
        return rc, None
    k = n.value
    return 0, {"pos": opos[:k].copy(), "off": ooff[:k + 1].copy(), "nbr": onbr[:ooff[k]].copy()}


def load_degenerate_pair():
    d = np.load(os.path.join(HERE, "golden", "degenerate_walk_bound_convex.npz"))
    mesh = {"pos": d["mesh_pos"], "off": d["mesh_off"], "nbr": d["mesh_nbr"]}
    conv = {"pos": d["conv_pos"], "off": d["conv_off"], "nbr": d["conv_nbr"]}
    return mesh, conv, d["planes"]


def check_degenerate_pair(E, oracle):
    """Both solids of the pair through clip_polyhedron: the engine's answer is the oracle's (the Convex comes back empty)."""
    mesh, conv, planes = load_degenerate_pair()
    eng = E.Engine(0)
    try:
        for s in (mesh, conv):
            ref = oracle.clip(s, planes)
            got = eng.clip_polyhedron(s, planes)
            assert_solid_equal(got, ref)
        assert oracle.clip(conv, planes)["pos"].shape[0] == 0
        # plane by plane as well: the clip that empties the Convex is the second
        cur_e, cur_o = conv, conv
        for k in range(planes.shape[0]):
            cur_o = oracle.clip(cur_o, planes[k:k + 1])
            cur_e = eng.clip_polyhedron(cur_e, planes[k:k + 1])
            assert_solid_equal(cur_e, cur_o)
            if cur_o["pos"].shape[0] == 0:
                break
    finally:
        eng.close()


def test_literal_equals_oracle_on_regular_solids(emul_lib_path, oracle):
    """Cells of a 12-cell pattern against the blob's Mesh and Convex, then random plane sets against the resulting fragments."""
    lib = _probe(emul_lib_path)
    rng = np.random.default_rng(5)
    sc = scenes.blob_scene(12)
    planes = oracle.place_cells(sc["v012"], sc["scale"], sc["translate"])
    fo = sc["face_off"]
    n_checked = 0
    frags = []
    for c in range(12):
        pl = planes[fo[c]:fo[c + 1]]
        for s in (sc["mesh"], sc["convex"]):
            ref = oracle.clip(s, pl)
            rc, got = _literal(lib, s, pl, cap_v=4096)
            assert rc == 0
            assert_solid_equal(got, ref)
            n_checked += 1
            if ref["pos"].shape[0]:
                frags.append(ref)
    for s in frags:
        ctr = s["pos"].mean(0)
        for _ in range(4):
            k = int(rng.integers(1, 7))
            nrm = rng.normal(size=(k, 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
            d = -(nrm @ ctr) + rng.uniform(-2.0, 8.0, size=k)
            pl = np.concatenate([nrm, d[:, None]], axis=1).astype(np.float32)
            ref = oracle.clip(s, pl)
            rc, got = _literal(lib, s, pl, cap_v=4096)
            assert rc == 0
            assert_solid_equal(got, ref)
            n_checked += 1
    assert n_checked >= 60


def test_literal_in_plane_vertices(emul_lib_path, oracle):
    """Planes through vertices, edges and faces of a cube (comp == 0 paths, :367-425 and the two-neighbour collapse)."""
    lib = _probe(emul_lib_path)
    s = oracle.unit_box()
    h = 0.5
    planes = [
        [1, 0, 0, 0], [1, 0, 0, h], [-1, 0, 0, h], [1, 1, 0, 0], [1, 1, 1, 0], [1, 1, 0, 2 * h], [-1, -1, 0, 2 * h],
        [1, 1, 1, 3 * h], [-1, -1, -1, 3 * h], [1, 1, 1, h], [-1, -1, -1, h], [0, 1, 0, -h], [0, -1, 0, -h],
    ]
    for p in planes:
        pl = np.asarray([p], np.float32)
        ref = oracle.clip(s, pl)
        rc, got = _literal(lib, s, pl)
        assert rc == 0
        assert_solid_equal(got, ref)
    for a in range(len(planes)):
        for b in range(a + 1, len(planes)):
            pl = np.asarray([planes[a], planes[b]], np.float32)
            ref = oracle.clip(s, pl)
            rc, got = _literal(lib, s, pl)
            assert rc == 0
            assert_solid_equal(got, ref)


def test_literal_degenerate_pair(emul_lib_path, oracle):
    lib = _probe(emul_lib_path)
    mesh, conv, planes = load_degenerate_pair()
    for s in (mesh, conv):
        ref = oracle.clip(s, planes)
        rc, got = _literal(lib, s, planes)
        assert rc == 0
        assert_solid_equal(got, ref)


def test_engine_degenerate_pair_emulation(emul_engine, oracle):
    check_degenerate_pair(emul_engine, oracle)


def test_engine_degenerate_pair_wide_variant(emul_engine_small, oracle):
    check_degenerate_pair(emul_engine_small, oracle)


def load_stale_id_pair():
    d = np.load(os.path.join(HERE, "golden", "degenerate_mesh_stale_id.npz"))
    mesh = {"pos": d["mesh_pos"], "off": d["mesh_off"], "nbr": d["mesh_nbr"]}
    conv = {"pos": d["conv_pos"], "off": d["conv_off"], "nbr": d["conv_nbr"]}
    return mesh, conv, d["planes"]


def check_stale_id_pair(E, oracle):
    """Refracture fuzz seed 90210, case 1271: a sliver Mesh of six vertices (three coincident pairs, every ring lists a
    neighbour twice).  Its clip leaves a link to a clipped vertex, which the reference renumbers through that vertex's ID
    of the plane before (Src/Poly.cpp:484-493) and carries on with four vertices, one ring [1, 2, 2, 2]: no polyhedron, an
    accident of what its memory held.  The degenerate policy: the engine flags it (SURTR_E_TOPOLOGY)
    instead of following the stale ID; the Convex of the pair, which is regular, is the oracle's."""
    from helpers import solid_has_doubled_neighbour
    mesh, conv, planes = load_stale_id_pair()
    eng = E.Engine(0)
    try:
        ref = oracle.clip(mesh, planes)
        assert ref["pos"].shape[0] == 4 and ref["nbr"][ref["off"][3]:ref["off"][4]].tolist() == [1, 2, 2, 2]
        assert solid_has_doubled_neighbour(ref)
        with pytest.raises(E.SurtrError) as e:
            eng.clip_polyhedron(mesh, planes)
        assert e.value.code == E.E_TOPOLOGY
        assert_solid_equal(eng.clip_polyhedron(conv, planes), oracle.clip(conv, planes))
    finally:
        eng.close()


def test_literal_stale_id_pair(emul_lib_path, oracle):
    lib = _probe(emul_lib_path)
    mesh, conv, planes = load_stale_id_pair()
    rc, got = _literal(lib, mesh, planes)
    assert rc == 2                                       # SURTR_E_TOPOLOGY: flagged, not emulated
    rc, got = _literal(lib, conv, planes)
    assert rc == 0
    assert_solid_equal(got, oracle.clip(conv, planes))


def test_engine_stale_id_pair_emulation(emul_engine, oracle):
    check_stale_id_pair(emul_engine, oracle)


def test_engine_stale_id_whole_event_emulation(emul_engine, oracle):
    """The whole event the pair comes from (96 first-level cells of a 54 x 150 torus, 3 cells per piece): the Mesh of that pair
    takes the literal clipper inside k_clip_pairs and the island split behind it."""
    from helpers import assert_event_equal_flagged
    from test_refracture import _refracture
    c, got, ref, npieces = _refracture(emul_engine, oracle, 96, 3, 54, 150)
    assert c.status == 0 and c.n_failed >= 1 and len(got["flagged_pairs"]) >= 1      # the event goes on without that pair
    assert_event_equal_flagged(got, ref)


def check_sliver_convex_walk_bound(E, oracle):
    """Refracture fuzz seed 13579, case 84 (tests/golden/sliver_convex_walk_bound.npz): a seven-vertex Convex with doubled
    neighbours.  After the first plane one of the second plane's relink walks wanders among clipped vertices; the reference
    stops it after as many steps as the solid has vertices at that moment and ends with nothing, the parallel clipper -- whose
    slots still hold the vertices clipped before -- let it run on to a new vertex and returned a tetrahedron (one fragment too
    many in the event).  Solids with a doubled neighbour now take the literal clipper from the start."""
    d = np.load(os.path.join(HERE, "golden", "sliver_convex_walk_bound.npz"))
    mesh = {"pos": d["mesh_pos"], "off": d["mesh_off"], "nbr": d["mesh_nbr"]}
    conv = {"pos": d["conv_pos"], "off": d["conv_off"], "nbr": d["conv_nbr"]}
    planes = d["planes"]
    eng = E.Engine(0)
    try:
        for k in range(1, planes.shape[0] + 1):
            for s in (conv, mesh):
                assert_solid_equal(eng.clip_polyhedron(s, planes[:k]), oracle.clip(s, planes[:k]))
        assert oracle.clip(conv, planes[:2])["pos"].shape[0] == 0
        # as a piece of an event: the cell that the planes belong to yields no fragment
        c = eng.upload_pieces([mesh], [conv])
        eng.upload_planes(np.asarray([0, planes.shape[0]], np.uint32), planes)
        c = eng.fracture_event(0, 1, flags=3)
        assert c.status == 0 and c.n_frag == 0
    finally:
        eng.close()


def test_sliver_convex_walk_bound_emulation(emul_engine, oracle):
    check_sliver_convex_walk_bound(emul_engine, oracle)
