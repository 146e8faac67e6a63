"""CPU tier: the product kernels compiled as a single-lane emulation (tests/emul) against the oracle.
This checks kernel *logic* (ordering rules, in-plane fallback, band reduction); races need the GPU tier."""
import os

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

from helpers import RTOL, assert_event_equal, fragment, run_event
from surtr_amd import meshgen, scenes


@pytest.mark.parametrize("flags", [0, 1, 2, 3])
def test_cube_8_cells(emul_engine, oracle, flags):
    c, got, ref = run_event(emul_engine, oracle, scenes.cube_scene(8), flags)
    assert c.n_frag == 8
    assert_event_equal(got, ref, render=bool(flags & 2))


def test_blob_64_cells(emul_engine, oracle):
    c, got, ref = run_event(emul_engine, oracle, scenes.blob_scene(64), 3)
    assert c.n_frag == ref["frag_ids"].shape[0] > 40
    assert_event_equal(got, ref)
    assert np.array_equal(got["mesh_pos"], ref["mesh_pos"])       # same float program => bit-exact here


def test_blob_1024_cells_sample(emul_engine, oracle):
    c, got, ref = run_event(emul_engine, oracle, scenes.blob_scene(1024), 3, cells=160)
    assert_event_equal(got, ref)


def test_torus_sample(emul_engine, oracle):
    sc = scenes.make_scene(*__import__("surtr_amd.meshgen", fromlist=["x"]).bumpy_torus(100, 60), 256)
    c, got, ref = run_event(emul_engine, oracle, sc, 3, cells=48)
    assert c.n_frag > 0
    assert_event_equal(got, ref)


def test_islands_are_split_like_the_reference(emul_engine, oracle):
    # two disjoint cubes in one piece: every cell that meets both yields two islands
    from surtr_amd import engine, meshgen
    v, t = meshgen.cube(1.0)
    v2 = np.concatenate([v, v + np.float32([5, 0, 0])])
    t2 = np.concatenate([t, t + 8])
    sc = scenes.make_scene(v2, t2, 6)
    c, got, ref = run_event(emul_engine, oracle, sc, 3)
    assert got["frag_ids"][:, 2].max() >= 1
    assert_event_equal(got, ref)


def test_outside_mask_skips_pieces(emul_engine, oracle):
    sc = scenes.cube_scene(8)
    eng = emul_engine.Engine(0)
    eng.upload_pieces([sc["mesh"], sc["mesh"]], [sc["convex"], sc["convex"]])
    eng.upload_pattern(sc["face_off"], sc["v012"])
    eng.place_cells(sc["scale"], sc["translate"])
    c = eng.fracture_event(0, 8, outside=[1, 0], flags=3)
    got = eng.download()
    eng.close()
    planes = oracle.place_cells(sc["v012"], sc["scale"], sc["translate"])
    ref = oracle.event([sc["mesh"]] * 2, [sc["convex"]] * 2, sc["face_off"], planes, outside=[1, 0])
    assert np.all(got["frag_ids"][:, 1] == 1)
    assert_event_equal(got, ref)


def test_clip_polyhedron_kats(emul_engine, oracle):
    eng = emul_engine.Engine(0)
    box = oracle.unit_box()
    for planes in ([[1, 1, 1, 0]], [[1, 0, 0, -0.5]], [[1, 1, 0, 0]], [[1, 0, 0, 0], [0, 1, 0, 0], [1, 1, 1, -0.2]],
                   [[0, 0, 0, 0]], [[0, 0, 0, 1]], [[1, 0, 0, 0.25], [-1, 0, 0, 0.25]]):
        pl = np.array(planes, np.float32)
        a = eng.clip_polyhedron(box, pl)
        b = oracle.clip(box, pl)
        assert a["pos"].shape == b["pos"].shape, planes
        assert np.array_equal(a["off"], b["off"]) and np.array_equal(a["nbr"], b["nbr"]), planes
        assert np.array_equal(a["pos"], b["pos"]), planes
    assert eng.clip_polyhedron(box, np.array([[1, 0, 0, 1.5]], np.float32))["pos"].shape[0] == 0
    eng.close()


@settings(max_examples=40, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(st.lists(st.tuples(st.floats(-1, 1), st.floats(-1, 1), st.floats(-1, 1), st.floats(-0.6, 0.3)), min_size=1, max_size=10),
       st.booleans())
def test_random_planes_on_box_and_blob(emul_engine, oracle, planes, use_blob):
    pl = np.array(planes, np.float32)
    pl = pl[np.abs(pl[:, :3]).sum(1) > 1e-3]
    if pl.shape[0] == 0:
        return
    solid = scenes.blob_scene(8)["mesh"] if use_blob else oracle.unit_box()
    if use_blob:
        solid = dict(solid, pos=solid["pos"] / 70.0)
    eng = emul_engine.Engine(0)
    a = eng.clip_polyhedron(solid, pl)
    eng.close()
    b = oracle.clip(solid, pl)
    assert np.array_equal(a["off"], b["off"]) and np.array_equal(a["nbr"], b["nbr"])
    assert np.array_equal(a["pos"], b["pos"])


def test_axis_aligned_grid_of_cells_hits_in_plane_path(emul_engine, oracle):
    # cube mesh cut by cells whose walls pass exactly through mesh vertices and edges (comp == 0 everywhere)
    from surtr_amd import meshgen, engine
    v, t = meshgen.cube(1.0)
    mesh = engine.neighbors_from_mesh(v, t)
    planes, off = [], [0]
    for ix in (-1, 0):
        for iy in (-1, 0):
            x0, x1, y0, y1 = ix, ix + 1, iy, iy + 1
            planes += [[-1, 0, 0, x0], [1, 0, 0, -x1], [0, -1, 0, y0], [0, 1, 0, -y1], [0, 0, -1, -1], [0, 0, 1, -1]]
            off.append(len(planes))
    planes = np.array(planes, np.float32)
    eng = emul_engine.Engine(0)
    eng.upload_pieces([mesh], [scenes.box_solid([2, 2, 2], [0, 0, 0])])
    eng.upload_planes(off, planes)
    c = eng.fracture_event(0, 4, flags=3)
    got = eng.download()
    eng.close()
    ref = oracle.event([mesh], [scenes.box_solid([2, 2, 2], [0, 0, 0])], off, planes)
    assert c.n_frag == 4
    assert_event_equal(got, ref)
    vols = [oracle.moments(fragment(got, k))[0] for k in range(4)]
    assert np.allclose(vols, 2.0, atol=1e-5)


def test_upload_rejects_bad_topology(emul_engine):
    sc = scenes.cube_scene(8)
    bad = dict(sc["mesh"], nbr=sc["mesh"]["nbr"].copy())
    bad["nbr"][0] = (bad["nbr"][0] + 3) % 8
    eng = emul_engine.Engine(0)
    with pytest.raises(emul_engine.SurtrError) as e:
        eng.upload_pieces([bad], [sc["convex"]])
    assert e.value.code == emul_engine.E_TOPOLOGY
    with pytest.raises(emul_engine.SurtrError) as e:
        eng.fracture_event(0, 1)
    assert e.value.code == emul_engine.E_STATE
    eng.close()


def test_capacity_error_is_reported(emul_engine):
    sc = scenes.blob_scene(64)
    eng = emul_engine.Engine(0)
    eng.set_arena(64, 256, 256)
    eng.upload_pieces([sc["mesh"]], [sc["convex"]])
    eng.upload_pattern(sc["face_off"], sc["v012"])
    eng.place_cells(sc["scale"], sc["translate"])
    with pytest.raises(emul_engine.SurtrError) as e:
        eng.fracture_event(0, 64)
    assert e.value.code == emul_engine.E_CAPACITY
    eng.close()


def test_in_plane_vertices_with_band_reduction(emul_engine, oracle):
    # planes through rows of torus vertices: in-plane (comp == 0) vertices next to dropped regions
    from surtr_amd import engine, meshgen
    v, t = meshgen.bumpy_torus(60, 40)
    mesh = engine.neighbors_from_mesh(v, t)
    assert (np.abs(v[:, 2]) < 1e-10).sum() > 50
    eng = emul_engine.Engine(0)
    for planes in ([[0, 0, 1, 0]], [[0, 0, -1, 0], [1, 0, 0, 0]], [[0, 1, 0, 0], [0, 0, 1, 0], [1, 0, 0, -0.5]],
                   [[1, 0, 0, -0.2], [0, 0, 1, 0], [0, 0, -1, 0]], [[0, 0, 1, 0], [0, 0, 1, 0]]):
        pl = np.array(planes, np.float32)
        a, b = eng.clip_polyhedron(mesh, pl), oracle.clip(mesh, pl)
        assert np.array_equal(a["off"], b["off"]) and np.array_equal(a["nbr"], b["nbr"]), planes
        assert np.array_equal(a["pos"], b["pos"]), planes
    eng.close()


@pytest.mark.parametrize("which", ["cube", "blob", "islands"])
def test_wide_global_variant(emul_engine_small, oracle, which):
    """Same kernels, LDS capacities shrunk so the 32-bit global-scratch Topo is exercised."""
    if which == "cube":
        sc, cells = scenes.cube_scene(8), None
    elif which == "blob":
        sc, cells = scenes.blob_scene(64), None
    else:
        from surtr_amd import meshgen
        v, t = meshgen.cube(1.0)
        sc, cells = scenes.make_scene(np.concatenate([v, v + np.float32([5, 0, 0])]), np.concatenate([t, t + 8]), 6), None
    c, got, ref = run_event(emul_engine_small, oracle, sc, 3, cells=cells)
    assert_event_equal(got, ref)
    assert np.array_equal(got["mesh_pos"], ref["mesh_pos"])


def test_ach_convex_matches_oracle(emul_engine, oracle):
    """PrepareFracture steps 1-6: limited hull normals, k-DOP planes (host helpers) and the GPU clip of the 2x box."""
    from surtr_amd import meshgen, engine
    for v, t in (meshgen.cube(), meshgen.blob(scale=70.0), meshgen.bumpy_torus(60, 40)):
        for limit in (4, 9, 20):
            assert np.array_equal(engine.hull_normals(v, limit), oracle.hull_normals(v, limit))
        eng = emul_engine.Engine(0)
        conv, planes = scenes.ach_convex(eng, v)
        eng.close()
        ext = v.max(0) - v.min(0)
        cen = ((v.max(0).astype(np.float64) + v.min(0).astype(np.float64)) / 2.0).astype(np.float32)
        n = oracle.hull_normals(v, 20)
        pl = oracle.kdop_planes(v, n, ach=True, max_axis_scale=float(ext.max()), gap_inv=2000.0)
        assert np.array_equal(planes, pl)
        ref = oracle.clip(scenes.box_solid(ext, cen), pl)
        assert np.array_equal(conv["off"], ref["off"]) and np.array_equal(conv["nbr"], ref["nbr"])
        assert np.array_equal(conv["pos"], ref["pos"])
        assert conv["pos"].shape[0] >= 8


def test_event_with_ach_convex(emul_engine, oracle):
    from surtr_amd import meshgen
    v, t = meshgen.blob(scale=70.0)
    sc = scenes.make_scene(v, t, 64)
    eng = emul_engine.Engine(0)
    sc["convex"], _ = scenes.ach_convex(eng, v)
    eng.close()
    c, got, ref = run_event(emul_engine, oracle, sc, 3, cells=48)
    assert_event_equal(got, ref)


def _oracle_prepare(oracle, v, t, n_cells, seed):
    """PrepareFracture restated with the oracle's pieces only (Src/Surtr.cpp:1747-1827)."""
    lo, hi = v.min(0), v.max(0)
    ext = (hi - lo).astype(np.float32)
    ctr = ((hi.astype(np.float64) + lo.astype(np.float64)) / 2.0).astype(np.float32)
    nrm = oracle.hull_normals(v, 20)
    pl = oracle.kdop_planes(v, nrm, ach=True, max_axis_scale=float(max(float(hi[a]) - float(lo[a]) for a in range(3))), gap_inv=2000.0)
    ach = oracle.clip(scenes.box_solid(ext, ctr), pl)
    mesh = oracle.neighbours_from_mesh(v, t)
    cells = oracle.voronoi_cells(oracle.seeds(n_cells, seed))
    return ach, mesh, cells, ext, ctr


def test_prepare_fracture_end_to_end(emul_engine, oracle):
    """Row f2: raw mesh in, initial compound out, against the oracle restatement of every step."""
    v, t = meshgen.blob(2, scale=3.0)
    eng = emul_engine.Engine(0)
    sc, c, got = scenes.prepare_fracture(eng, v, t, n_cells=12, seed=4711)
    ach, mesh, cells, ext, ctr = _oracle_prepare(oracle, v, t, 12, 4711)
    assert np.array_equal(sc["convex"]["nbr"], ach["nbr"]) and np.allclose(sc["convex"]["pos"], ach["pos"], rtol=1e-5, atol=1e-6)
    assert np.array_equal(sc["mesh"]["nbr"], mesh["nbr"])
    planes = oracle.place_cells(sc["v012"], ext, ctr)
    ref = oracle.event([mesh], [ach], sc["face_off"], planes, refit=True, render=True, threads=2)
    assert c.status == 0 and c.n_frag >= 8
    assert_event_equal(got, ref)
    meshes, convexes = scenes.fragments_as_pieces(got)
    assert len(meshes) == c.n_frag == len(convexes)
    eng.close()


def test_image_arena_exhaustion_falls_back(emul_engine, oracle, monkeypatch):
    """k_prep_pairs leaves the pairs it has no image room for to k_clip_pairs' own pre-pass: same result."""
    monkeypatch.setenv("SURTR_IMG_BYTES", "16384")
    sc = scenes.make_scene(*meshgen.bumpy_torus(40, 24), 24)
    c, got, ref = run_event(emul_engine, oracle, sc)
    assert c.status == 0 and c.n_frag > 10
    assert_event_equal(got, ref)


def test_degenerate_ach_of_a_box_is_refused(emul_engine):
    """Found by scripts/fuzz_gpu.py: the ACH of an axis-aligned box has slabs that coincide with the box faces up to
    rounding and near-duplicate vertices; a cell plane then leaves a clipped vertex linked from a surviving one, where the
    reference indexes with ID = -1 (Src/Poly.cpp:464-495).  The engine must flag the pair (SURTR_E_TOPOLOGY in its status), not crash."""
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "degenerate_ach_cube.npz"))
    eng = emul_engine.Engine(0)
    eng.upload_pieces([{"pos": d["mesh_pos"], "off": d["mesh_off"], "nbr": d["mesh_nbr"]}],
                      [{"pos": d["conv_pos"], "off": d["conv_off"], "nbr": d["conv_nbr"]}])
    eng.upload_planes(np.uint32([0, d["planes"].shape[0]]), d["planes"])
    # the degenerate policy: the pair is flagged and yields no fragment, the event goes on
    c = eng.fracture_event(0, 1, flags=3)
    assert c.status == 0 and c.n_frag == 0 and c.n_failed == 1
    assert eng.pair_status(1).tolist() == [emul_engine.E_TOPOLOGY]
    # the context stays usable
    sc = scenes.cube_scene(8)
    eng.upload_pieces([sc["mesh"]], [sc["convex"]]); eng.upload_pattern(sc["face_off"], sc["v012"]); eng.place_cells(sc["scale"], sc["translate"])
    assert eng.fracture_event(0, 8).n_frag == 8
    eng.close()


@pytest.mark.parametrize("which,tiny", [("cube", False), ("blob", False), ("torus", False), ("torus12", True), ("torus16", True)])
def test_half_size_kernel_and_its_retry_list(emul_lib_path, oracle, monkeypatch, which, tiny):
    """k_clip_pairs_half (light pairs, half-size LDS topology) forced on: same event.  In the tiny-capacity build it admits
    solids with no room to grow, so pairs outgrow it and are redone by the retry launch of k_clip_pairs."""
    monkeypatch.setenv("SURTR_HALF", "1")
    sc = {"cube": lambda: scenes.cube_scene(8), "blob": lambda: scenes.blob_scene(64),
          "torus": lambda: scenes.make_scene(*meshgen.bumpy_torus(40, 24), 24),
          "torus12": lambda: scenes.make_scene(*meshgen.bumpy_torus(12, 8), 6),
          "torus16": lambda: scenes.make_scene(*meshgen.bumpy_torus(16, 10), 12)}[which]()
    cells = sc["n_cells"]
    from surtr_amd import engine as E
    E._use_library_for_tests(os.path.join(os.path.dirname(emul_lib_path), "libsurtr_emul_small.so" if tiny else "libsurtr_emul.so"))
    try:
        eng = E.Engine(0)
        eng.upload_pieces([sc["mesh"]], [sc["convex"]]); eng.upload_pattern(sc["face_off"], sc["v012"]); eng.place_cells(sc["scale"], sc["translate"])
        c = eng.fracture_event(0, cells, flags=3)
        got = eng.download()
        qs = eng.queue_stats()
        eng.close()
    finally:
        E._use_library_for_tests(None)
    planes = oracle.place_cells(sc["v012"], sc["scale"], sc["translate"])
    ref = oracle.event([sc["mesh"]], [sc["convex"]], sc["face_off"], planes, refit=True, render=True, threads=4, cell_end=cells)
    assert c.status == 0
    assert_event_equal(got, ref)
    assert qs[65:71].sum() > 0, "no pair went to the half-size kernel"
    if tiny:
        assert qs[64] > 0, "the retry list was not exercised"


def test_retry_after_the_half_size_kernel_squeezed(emul_lib_path, oracle, monkeypatch):
    """Pairs that outgrow the half-size topology after a few planes (and a squeeze) are redone from the image by the
    retry launch: the image must still be what k_prep_pairs wrote (the half-size kernel works on a copy of its positions;
    the other kernels use them in place).  Random small tori in a medium-capacity build; caught a real bug."""
    monkeypatch.setenv("SURTR_HALF", "1")
    from surtr_amd import engine as E
    E._use_library_for_tests(os.path.join(os.path.dirname(emul_lib_path), "libsurtr_emul_mid.so"))
    rng = np.random.default_rng(5)
    retried = 0
    try:
        for case in range(16):
            nu, nv, cells = int(rng.integers(10, 40)), int(rng.integers(8, 24)), int(rng.choice([4, 6, 9, 14]))
            sc = scenes.make_scene(*meshgen.bumpy_torus(nu, nv), cells, seeds=scenes.uniform_seeds(cells, int(rng.integers(1, 1 << 30))))
            eng = E.Engine(0)
            eng.upload_pieces([sc["mesh"]], [sc["convex"]]); eng.upload_pattern(sc["face_off"], sc["v012"]); eng.place_cells(sc["scale"], sc["translate"])
            c = eng.fracture_event(0, cells, flags=3)
            got = eng.download()
            retried += int(eng.queue_stats()[64])
            eng.close()
            planes = oracle.place_cells(sc["v012"], sc["scale"], sc["translate"])
            ref = oracle.event([sc["mesh"]], [sc["convex"]], sc["face_off"], planes, refit=True, render=True, threads=4, cell_end=cells)
            assert c.status == 0
            assert_event_equal(got, ref)
    finally:
        E._use_library_for_tests(None)
    assert retried > 5


def test_sliver_piece_whose_convex_clip_is_invalid_in_the_reference(emul_engine, oracle):
    """Found by scripts/fuzz_refracture_gpu.py: a first-level fragment of four vertices, two pairs of them coincident; its
    refitted Convex has coincident vertices too.  A cell plane of the second level leaves the reference's clip of that
    Convex with a ring entry that points past the last vertex (the restatement counts it, orc_links_off_the_array, and stops
    with an empty solid), the reference carries on, nothing is left of the Mesh, the pair yields no fragment.  The engine does the same:
    the inconsistent Convex only fails the event (SURTR_E_TOPOLOGY) if a fragment would have to carry it."""
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "degenerate_sliver_convex.npz"))
    mesh = {"pos": d["mesh_pos"], "off": d["mesh_off"], "nbr": d["mesh_nbr"]}
    conv = {"pos": d["conv_pos"], "off": d["conv_off"], "nbr": d["conv_nbr"]}
    oracle.links_off_the_array(reset=True)
    ref = oracle.clip(conv, d["planes"])
    # the reference's result is not a solid: its compaction stores a link that names no vertex (the restatement stops there)
    assert oracle.links_off_the_array(reset=True) > 0 and ref["pos"].shape[0] == 0
    fo = np.uint32([0, d["planes"].shape[0]])
    ev = oracle.event([mesh], [conv], fo, d["planes"], refit=True, render=True, threads=1)
    assert ev["frag_ids"].shape[0] == 0
    eng = emul_engine.Engine(0)
    eng.upload_pieces([mesh], [conv])
    eng.upload_planes(fo, d["planes"])
    c = eng.fracture_event(0, 1, flags=3)
    assert c.status == 0 and c.n_frag == 0
    eng.close()


def test_fragment_with_a_face_loop_through_one_vertex_twice(emul_engine, oracle):
    """Found by scripts/fuzz_refracture_gpu.py: a degenerate second-level fragment (coincident vertices) whose half-edge loop
    passes through a vertex twice although no ring lists a neighbour twice.  The reference closes a face when its walk is back
    at the start vertex (Src/Poly.cpp:100-118), so that loop is two faces (a triangle and a pentagon here); k_faces used to
    emit one and drop four triangles.  Such fragments are now redone by the literal path."""
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pinched_face_fragment.npz"))
    mesh = {"pos": d["mesh_pos"], "off": d["mesh_off"], "nbr": d["mesh_nbr"]}
    conv = {"pos": d["conv_pos"], "off": d["conv_off"], "nbr": d["conv_nbr"]}
    fo = d["fo"].astype(np.uint32)
    eng = emul_engine.Engine(0)
    eng.upload_pieces([mesh], [conv]); eng.upload_planes(fo, d["planes"])
    c = eng.fracture_event(0, len(fo) - 1, flags=3)
    got = eng.download()
    eng.close()
    ref = oracle.event([mesh], [conv], fo, d["planes"], refit=True, render=True, threads=2)
    assert c.status == 0
    assert_event_equal(got, ref)


def check_nonterminating_faces_fragment(E, oracle):
    """Seed 27182, case 264 of scripts/fuzz_refracture_gpu.py: one fragment (125 vertices, two rings with a doubled
    neighbour) on which the reference's ExtractFaces never ends (Src/Poly.cpp:100-118).  It is refused alone -- no
    triangles, SURTR_E_TOPOLOGY in frag_status, counts.n_failed = 1 -- and the event is SURTR_OK with every other fragment
    equal to the oracle's (whose restatement stops such a walk after H steps)."""
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "nonterminating_faces_fragment.npz"))
    mesh = {"pos": d["mesh_pos"], "off": d["mesh_off"], "nbr": d["mesh_nbr"]}
    conv = {"pos": d["conv_pos"], "off": d["conv_off"], "nbr": d["conv_nbr"]}
    fo = d["fo"].astype(np.uint32)
    eng = E.Engine(0)
    try:
        eng.upload_pieces([mesh], [conv]); eng.upload_planes(fo, d["planes"])
        c = eng.fracture_event(0, len(fo) - 1, flags=3)
        got = eng.download()
    finally:
        eng.close()
    ref = oracle.event([mesh], [conv], fo, d["planes"], refit=True, render=True, threads=2)
    assert c.status == 0 and c.n_failed == 1 and c.n_frag >= 4
    bad = np.nonzero(got["frag_status"])[0]
    assert bad.shape[0] == 1 and got["frag_status"][bad[0]] == E.E_TOPOLOGY
    assert got["idx_off"][bad[0]] == got["idx_off"][bad[0] + 1]                      # no triangles for that one
    assert got["mesh_vert_off"][bad[0] + 1] - got["mesh_vert_off"][bad[0]] == 125
    assert_event_equal(got, ref)
    # the single-solid operators refuse that solid by itself
    from helpers import fragment
    eng = E.Engine(0)
    try:
        with pytest.raises(E.SurtrError) as e:
            eng.extract_faces(fragment(got, int(bad[0])))
        assert e.value.code == E.E_TOPOLOGY
    finally:
        eng.close()


def test_nonterminating_faces_fragment_is_isolated(emul_engine, oracle):
    check_nonterminating_faces_fragment(emul_engine, oracle)


def test_deep_lobed_mesh_emulated(emul_engine, oracle):
    """meshgen.urchin x 64 cells (the GPU tier also runs 1 024): several islands per cell in a quarter of the cells."""
    c, got, ref = run_event(emul_engine, oracle, scenes.urchin_scene(64))
    assert c.status == 0 and int(got["frag_ids"][:, 2].max()) >= 1
    assert_event_equal(got, ref)


def test_faces_second_tier_and_workgroup_budget(emul_engine, oracle, monkeypatch):
    """Pieces of some 100 000 vertices: the regular k_faces workgroups get scratch for fragments of up to 2^19 half-edges and
    hand larger ones to a second launch with full-size scratch; the persistent kernels run with as many workgroups as a share
    of the device memory holds.  Forced here with a 256-half-edge tier and a 64 MB "device": same event."""
    monkeypatch.setenv("SURTR_FACES_TIER_HE", "256")
    monkeypatch.setenv("SURTR_MEM_BUDGET_MB", "64")
    c, got, ref = run_event(emul_engine, oracle, scenes.blob_scene(64), 3)
    assert c.status == 0 and c.n_frag == ref["frag_ids"].shape[0] > 40
    assert_event_equal(got, ref)
    eng = emul_engine.Engine(0)
    try:
        # the single-solid face extraction goes through the same two launches
        sc = scenes.blob_scene(8)
        fo, fi = eng.extract_faces(sc["mesh"])
        rfo, rfi = oracle.extract_faces(sc["mesh"])
        assert np.array_equal(fo, rfo) and np.array_equal(fi, rfi)
    finally:
        eng.close()


def check_refit_invalid_in_reference(E, oracle):
    """Refracture fuzz seed 555002, case 82, fragment 536 (tests/golden/refit_invalid_in_reference.npz: Mesh and Convex of nine
    vertices each): the reference's refit leaves a link to a clipped vertex, renumbers it through that vertex's stale ID
    (Src/Poly.cpp:484-493) and carries on with a Convex of seven vertices that has a one-way link.  The engine flags the
    fragment instead of following the stale ID (it keeps its un-refitted Convex), event SURTR_OK."""
    from helpers import solid_is_polyhedron
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "refit_invalid_in_reference.npz"))
    mesh = {"pos": d["mesh_pos"], "off": d["mesh_off"], "nbr": d["mesh_nbr"]}
    conv = {"pos": d["conv_pos"], "off": d["conv_off"], "nbr": d["conv_nbr"]}
    assert solid_is_polyhedron(mesh) and solid_is_polyhedron(conv)
    ref = oracle.refit(conv, mesh, 4)
    assert ref["pos"].shape[0] == 7 and not solid_is_polyhedron(ref)
    cube = scenes.cube_scene(8)
    eng = E.Engine(0)
    try:
        # asked for that one solid, the engine says that there is no valid answer (the degenerate policy, DESIGN section 3.7:
        # the stale ID is not followed)
        with pytest.raises(E.SurtrError) as e:
            eng.refit_solid(mesh, conv)
        assert e.value.code == E.E_TOPOLOGY
        # among other fragments: flagged, keeps the Convex it had, the others are refitted
        eng.load_fragments([cube["mesh"], mesh, cube["mesh"]], [cube["convex"], conv, cube["convex"]])
        eng.event_refit()
        c = eng.event_counts()
        ev = eng.download()
        assert c.status == 0 and c.n_failed == 1 and ev["frag_status"].tolist() == [0, E.E_TOPOLOGY, 0]
        r1 = fragment(ev, 1, "conv")
        assert np.array_equal(r1["off"], conv["off"]) and np.array_equal(r1["nbr"], conv["nbr"]) and np.array_equal(r1["pos"], conv["pos"])
        ref0 = oracle.refit(cube["convex"], cube["mesh"], 4)
        for k in (0, 2):
            r = fragment(ev, k, "conv")
            assert np.array_equal(r["off"], ref0["off"]) and np.array_equal(r["nbr"], ref0["nbr"])
        # the flagged fragment goes on through the triangulation of the event (its Mesh is what is triangulated)
        eng.event_triangulate()
        assert eng.event_counts().status == 0
    finally:
        eng.close()


def test_refit_result_that_is_no_polyhedron(emul_engine, oracle):
    check_refit_invalid_in_reference(emul_engine, oracle)
