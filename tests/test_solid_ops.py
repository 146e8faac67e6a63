"""The per-Piece tasks for ONE solid (surtr_refit_solid / surtr_extract_faces / surtr_triangulate), surtr_load_fragments,
surtr_transform_pieces and surtr_pieces_from_event against the oracle -- kernel logic on the single-lane emulation here,
the same cases on the MI355X in tests/test_gpu_parity.py (they share `cases_*` below)."""
import numpy as np
import pytest

from helpers import RTOL, assert_event_equal, fragment
from surtr_amd import scenes


def _blob_event(oracle, n_cells=12):
    sc = scenes.blob_scene(n_cells)
    planes = oracle.place_cells(sc["v012"], sc["scale"], sc["translate"])
    ev = oracle.event([sc["mesh"]], [sc["convex"]], sc["face_off"], planes, refit=False, render=False, threads=4)
    return sc, planes, ev


def check_single_solid_ops(E, oracle):
    sc, planes, ev = _blob_event(oracle)
    eng = E.Engine(0)
    try:
        for k in range(0, ev["frag_ids"].shape[0], 3):
            mesh, conv = fragment(ev, k, "mesh"), fragment(ev, k, "conv")
            # m_refittingTask
            got = eng.refit_solid(mesh, conv)
            ref = oracle.refit(conv, mesh, 4)
            assert np.array_equal(got["off"], ref["off"]) and np.array_equal(got["nbr"], ref["nbr"])
            assert np.allclose(got["pos"], ref["pos"], rtol=RTOL, atol=1e-6)
            # ExtractFaces
            fo, fi = eng.extract_faces(mesh)
            rfo, rfi = oracle.extract_faces(mesh)
            assert np.array_equal(fo, rfo) and np.array_equal(fi, rfi)
            # RenderPolyhedron, both branches, custom colour
            for convex_flag, solid in ((False, mesh), (True, conv)):
                vnc, idx = eng.triangulate(solid, is_convex=convex_flag, color=(0.5, 0.25, 1.0))
                rv, ri = oracle.render(solid, convex=convex_flag, colour=(0.5, 0.25, 1.0))
                assert np.array_equal(idx, ri), (k, convex_flag)
                assert np.array_equal(vnc, rv)
        # the unit box (Poly::GetBB): 6 quads -> 12 fan triangles
        vnc, idx = eng.triangulate(oracle.unit_box(), is_convex=True)
        assert idx.shape[0] == 36 and np.all(vnc[:, 6:] == 0.25)
    finally:
        eng.close()


def check_load_fragments_then_refit_and_triangulate(E, oracle):
    sc, planes, ev = _blob_event(oracle)
    n = ev["frag_ids"].shape[0]
    meshes = [fragment(ev, k, "mesh") for k in range(n)]
    convs = [fragment(ev, k, "conv") for k in range(n)]
    ref = oracle.event([sc["mesh"]], [sc["convex"]], sc["face_off"], planes, refit=True, render=True, threads=4)
    eng = E.Engine(0)
    try:
        eng.load_fragments(meshes, convs, ev["frag_ids"])
        eng.event_refit()
        eng.event_triangulate(False)
        got = eng.download()
    finally:
        eng.close()
    assert_event_equal(got, ref)
    assert not got["frag_status"].any()


def _world_matrices(n, seed=3):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        a = rng.normal(size=(3, 3)); q, _ = np.linalg.qr(a)
        if np.linalg.det(q) < 0:
            q[:, 0] = -q[:, 0]
        w = np.eye(4); w[:3, :3] = q * rng.uniform(0.8, 1.25); w[:3, 3] = rng.uniform(-2, 2, 3)
        out.append(w)
    return np.asarray(out, np.float32)


def check_transform_pieces(E, oracle):
    """ExecuteFractureRoutine's pre-transform (Src/Surtr.cpp:1846-1851): pieces moved on the device, then an event."""
    sc, planes, ev = _blob_event(oracle, 6)
    n = ev["frag_ids"].shape[0]
    meshes = [fragment(ev, k, "mesh") for k in range(n)]
    convs = [fragment(ev, k, "conv") for k in range(n)]
    W = _world_matrices(n)
    m2 = [oracle.transform(m, W[k]) for k, m in enumerate(meshes)]
    c2 = [oracle.transform(c, W[k]) for k, c in enumerate(convs)]
    # one pattern over the box of everything after the move
    allp = np.concatenate([m["pos"] for m in m2])
    lo, hi = allp.min(0), allp.max(0)
    scale, shift = (hi - lo).astype(np.float32), ((hi.astype(np.float64) + lo) / 2).astype(np.float32)
    pl = oracle.place_cells(sc["v012"], scale, shift)
    ref = oracle.event(m2, c2, sc["face_off"], pl, refit=True, render=True, threads=4)
    eng = E.Engine(0)
    try:
        eng.upload_pieces(meshes, convs)
        eng.transform_pieces(W)
        eng.upload_pattern(sc["face_off"], sc["v012"])
        eng.place_cells(scale, shift)
        eng.fracture_event(0, sc["n_cells"])
        got = eng.download()
        # a second transform in steady state allocates nothing
        eng.transform_pieces(_world_matrices(n, 5))
        assert eng.upload_stats()[1] == 0
    finally:
        eng.close()
    assert_event_equal(got, ref)


def check_pieces_from_event(E, oracle):
    """Two-level refracture with the first level's fragments staying on the device (BASELINE configs[4] in small)."""
    sc, planes, ev = _blob_event(oracle, 10)
    ev1 = oracle.event([sc["mesh"]], [sc["convex"]], sc["face_off"], planes, refit=True, render=False, threads=4)
    n1 = ev1["frag_ids"].shape[0]
    keep = np.ones(n1, np.uint8); keep[1::4] = 0
    kept = [k for k in range(n1) if keep[k]]
    meshes = [fragment(ev1, k, "mesh") for k in kept]
    convs = [fragment(ev1, k, "conv") for k in kept]
    rs = scenes.refracture_scene(meshes, convs, 5)
    eng = E.Engine(0)
    try:
        eng.upload_pieces([sc["mesh"]], [sc["convex"]])
        eng.upload_pattern(sc["face_off"], sc["v012"])
        eng.place_cells(sc["scale"], sc["translate"])
        eng.fracture_event(0, sc["n_cells"], flags=1)
        assert eng.pieces_from_event(keep) == len(kept)
        eng.upload_pattern(rs["face_off"], rs["v012"])
        eng.place_cells_groups(rs["group_cell_off"], rs["scales"], rs["shifts"])
        eng.fracture_pairs(rs["pair_cell"], rs["pair_piece"])
        got = eng.download()
    finally:
        eng.close()
    # the same chain with nothing but sizes crossing the bus: cells built on the device (one diagram per piece), placed over
    # the pieces' boxes taken on the device
    eng = E.Engine(0)
    try:
        eng.upload_pieces([sc["mesh"]], [sc["convex"]])
        eng.upload_pattern(sc["face_off"], sc["v012"])
        eng.place_cells(sc["scale"], sc["translate"])
        eng.fracture_event(0, sc["n_cells"], flags=1)
        n = eng.pieces_from_event(keep)
        seeds = np.concatenate([scenes.uniform_seeds(5, scenes.SEED + p) for p in range(n)])
        go = np.arange(0, 5 * n + 1, 5, dtype=np.uint32)
        eng.build_cells(seeds, go)
        eng.place_cells_in_pieces(go)
        eng.fracture_pairs(rs["pair_cell"], rs["pair_piece"])
        got2 = eng.download()
    finally:
        eng.close()
    for k in got:
        assert np.array_equal(got[k], got2[k]), k
    planes2 = np.concatenate([oracle.place_cells(rs["v012"][rs["face_off"][rs["group_cell_off"][g]]:rs["face_off"][rs["group_cell_off"][g + 1]]],
                                                 rs["scales"][g], rs["shifts"][g]) for g in range(len(kept))])
    parts = []
    for g in range(len(kept)):
        c0, c1 = int(rs["group_cell_off"][g]), int(rs["group_cell_off"][g + 1])
        fo = rs["face_off"][c0:c1 + 1] - rs["face_off"][c0]
        pl = planes2[rs["face_off"][c0]:rs["face_off"][c1]]
        r = oracle.event([meshes[g]], [convs[g]], fo, pl, refit=True, render=True, threads=2)
        r["frag_ids"][:, 0] += c0; r["frag_ids"][:, 1] = g
        parts.append(r)
    from surtr_amd import engine as eng_mod
    ref = eng_mod.merge_fragments(parts)
    assert_event_equal(got, ref)


CASES = [check_single_solid_ops, check_load_fragments_then_refit_and_triangulate, check_transform_pieces, check_pieces_from_event]


@pytest.mark.parametrize("case", CASES, ids=lambda f: f.__name__)
def test_solid_ops_emulated(emul_engine, oracle, case):
    case(emul_engine, oracle)


def test_bad_links_are_refused_by_the_device_check(emul_engine, oracle):
    box = oracle.unit_box()
    bad = dict(box, nbr=box["nbr"].copy())
    bad["nbr"][0] = 2          # vertex 0 now links 2, which does not link back
    eng = emul_engine.Engine(0)
    try:
        with pytest.raises(emul_engine.SurtrError) as e:
            eng.upload_pieces([bad], [box])
        assert e.value.code == emul_engine.E_TOPOLOGY
        eng.upload_pieces([box], [box])      # the context is usable afterwards
    finally:
        eng.close()
