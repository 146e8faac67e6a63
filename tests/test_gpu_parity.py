"""GPU tier (-m gpu): the HIP path, called through the C ABI, against the oracle on the same seeded inputs,
against the committed golden fixtures, and -- at BASELINE.json's full size -- through size-independent
properties.  Bit-exact for indices/topology; coordinates within 1e-5 relative (north_star)."""
import hashlib
import json
import os

import numpy as np
import pytest

from helpers import RTOL, assert_event_equal, fragment, run_event
from surtr_amd import meshgen, scenes

pytestmark = pytest.mark.gpu
HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOPO = ("frag_ids", "mesh_vert_off", "mesh_nbr_off", "mesh_nbr", "conv_vert_off", "conv_nbr_off", "conv_nbr", "idx_off", "idx")


@pytest.fixture(scope="module")
def torus_run():
    from surtr_amd import engine
    from oracle import oracle
    engine._use_library_for_tests(None)
    sc = scenes.torus_scene(4096)
    c, got, ref = run_event(engine, oracle, sc, 3, threads=16)
    return sc, c, got, ref


@pytest.mark.parametrize("flags", [0, 1, 2, 3])
def test_cube_8_cells(gpu_engine, oracle, flags):
    c, got, ref = run_event(gpu_engine, oracle, scenes.cube_scene(8), flags)
    assert c.n_frag == 8 and c.status == 0
    assert_event_equal(got, ref, render=bool(flags & 2))


def test_cube_golden_fixture(gpu_engine):
    g = np.load(os.path.join(HERE, "cube8.npz"))
    sc = scenes.cube_scene(8)
    eng = gpu_engine.Engine(0)
    eng.upload_pieces([sc["mesh"]], [sc["convex"]])
    eng.upload_planes(g["face_off"], g["planes"])
    eng.fracture_event(0, 8, flags=3)
    got = eng.download()
    eng.close()
    for k in TOPO:
        assert np.array_equal(got[k], g["out_" + k]), k
    for k in ("mesh_pos", "conv_pos"):
        assert np.allclose(got[k], g["out_" + k], rtol=RTOL, atol=1e-6), k


def test_place_cells_matches_oracle(gpu_engine, oracle):
    # cell placement kernel (A3): compare through a clip that is sensitive to the planes
    sc = scenes.blob_scene(64)
    planes = oracle.place_cells(sc["v012"], sc["scale"], sc["translate"])
    a = gpu_engine.Engine(0)
    a.upload_pieces([sc["mesh"]], [sc["convex"]])
    a.upload_pattern(sc["face_off"], sc["v012"])
    a.place_cells(sc["scale"], sc["translate"])
    a.fracture_event(0, 64, flags=0)
    x = a.download()
    a.upload_planes(sc["face_off"], planes)
    a.fracture_event(0, 64, flags=0)
    y = a.download()
    a.close()
    for k in ("mesh_pos", "mesh_nbr", "conv_pos", "conv_nbr"):
        assert np.array_equal(x[k], y[k]), k


@pytest.mark.parametrize("name,n", [("blob64", 64), ("blob1024", 1024)])
def test_blob_configs(gpu_engine, oracle, name, n):
    c, got, ref = run_event(gpu_engine, oracle, scenes.blob_scene(n), 3, threads=8)
    assert c.status == 0
    assert_event_equal(got, ref)
    want = json.load(open(os.path.join(HERE, "digests.json")))[name]
    assert c.n_frag == want["n_frag"] and c.mesh_verts == want["mesh_verts"] and c.n_idx == want["n_idx"]
    for k in TOPO:
        assert hashlib.sha256(np.ascontiguousarray(got[k]).tobytes()).hexdigest() == want[k], k
    assert abs(float(got["mesh_pos"].astype(np.float64).sum()) - want["mesh_pos_sum"]) <= 1e-5 * abs(want["mesh_pos_sum"]) + 1e-3


@pytest.mark.parametrize("which", ["blob1024", "torus512"])
def test_half_size_kernel_forced_on(gpu_engine, oracle, monkeypatch, which):
    """k_clip_pairs_half is chosen per upload from the piece sizes (small pieces: refracture); forced on here for large
    pieces too: light pairs take the half-size LDS topology, the others the regular kernel beside it, same event."""
    monkeypatch.setenv("SURTR_HALF", "1")
    sc = scenes.blob_scene(1024) if which == "blob1024" else scenes.torus_scene(4096)
    cells = 1024 if which == "blob1024" else 512
    eng = gpu_engine.Engine(0)
    eng.upload_pieces([sc["mesh"]], [sc["convex"]]); eng.upload_pattern(sc["face_off"], sc["v012"]); eng.place_cells(sc["scale"], sc["translate"])
    c = eng.fracture_event(0, cells, flags=3)
    got = eng.download()
    qs = eng.queue_stats()
    eng.close()
    planes = oracle.place_cells(sc["v012"], sc["scale"], sc["translate"])
    ref = oracle.event([sc["mesh"]], [sc["convex"]], sc["face_off"], planes, refit=True, render=True, threads=16, cell_end=cells)
    assert c.status == 0
    assert_event_equal(got, ref)
    assert qs[65:71].sum() > 0, "no pair went to the half-size kernel"


def test_torus_4096_full_event_vs_oracle_and_digest(torus_run):
    sc, c, got, ref = torus_run
    assert c.status == 0 and c.n_pairs == 4096
    assert_event_equal(got, ref)
    want = json.load(open(os.path.join(HERE, "digests.json")))["torus4096"]
    assert c.n_frag == want["n_frag"] and c.mesh_verts == want["mesh_verts"] and c.n_idx == want["n_idx"]
    for k in TOPO:
        assert hashlib.sha256(np.ascontiguousarray(got[k]).tobytes()).hexdigest() == want[k], k


def test_torus_4096_properties(torus_run, oracle):
    sc, c, got, ref = torus_run
    nf = c.n_frag
    # (1) fragments partition the solid: sum of volumes = volume of the mesh
    vols = np.array([oracle.moments(fragment(got, k))[0] for k in range(nf)])
    whole = oracle.moments(sc["mesh"])[0]
    # float32 clipping of 4096 cells: the oracle (bit-identical output) shows the same 2e-5 drift
    assert abs(vols.sum() - whole) / whole < 1e-4
    planes = oracle.place_cells(sc["v012"], sc["scale"], sc["translate"])
    mvo, mno = got["mesh_vert_off"].astype(np.int64), got["mesh_nbr_off"].astype(np.int64)
    ioff = got["idx_off"].astype(np.int64)
    for k in range(nf):
        fr = fragment(got, k)
        nv = fr["pos"].shape[0]
        off = fr["off"].astype(np.int64)
        # (2) neighbour links are symmetric (Src/Poly.cpp:253-260)
        src = np.repeat(np.arange(nv), np.diff(off))
        pairs = set(zip(src.tolist(), fr["nbr"].tolist()))
        assert all((b, a) in pairs for a, b in pairs)
        # (3) every vertex satisfies every plane of its cell
        cell = int(got["frag_ids"][k, 0])
        pl = planes[sc["face_off"][cell]:sc["face_off"][cell + 1]].astype(np.float64)
        s = fr["pos"].astype(np.float64) @ pl[:, :3].T + pl[:, 3]
        assert s.max() < 1e-4
        # (4) index buffers reference only the fragment's own vertices, whole triangles
        idx = got["idx"][ioff[k]:ioff[k + 1]]
        assert idx.size % 3 == 0 and (idx.size == 0 or idx.max() < nv)
        # (5) every half-edge lies in exactly one face loop; triangle count = sum(len-2) unless a face was dropped
        fo, fi = oracle.extract_faces(fr)
        dup = any(len(set(fr["nbr"][off[v]:off[v + 1]].tolist())) != off[v + 1] - off[v] for v in range(nv))
        if dup:      # sliver fragments may list a neighbour twice; the reference's visited-pair set then skips an edge
            assert fo[-1] <= fr["nbr"].shape[0]
        else:
            assert fo[-1] == fr["nbr"].shape[0]
        assert idx.size <= 3 * int((np.diff(fo.astype(np.int64)) - 2).sum())
    # (6) cell-major order, islands numbered from 0
    ids = got["frag_ids"]
    key = ids[:, 0].astype(np.int64) * 10 ** 6 + ids[:, 1] * 10 ** 3 + ids[:, 2]
    assert np.all(np.diff(key) > 0)


def test_clip_polyhedron_operator(gpu_engine, oracle):
    eng = gpu_engine.Engine(0)
    box = oracle.unit_box()
    for planes in ([[1, 1, 1, 0]], [[1, 0, 0, -0.5]], [[1, 1, 0, 0]], [[0, 0, 0, 0]], [[1, 0, 0, 0.25], [-1, 0, 0, 0.25]]):
        pl = np.array(planes, np.float32)
        a, b = eng.clip_polyhedron(box, pl), oracle.clip(box, pl)
        assert np.array_equal(a["off"], b["off"]) and np.array_equal(a["nbr"], b["nbr"]), planes
        assert np.allclose(a["pos"], b["pos"], rtol=RTOL, atol=1e-7), planes
    sc = scenes.blob_scene(8)
    rng = np.random.default_rng(5)
    for _ in range(6):
        n = rng.normal(size=(5, 3)).astype(np.float32)
        pl = np.concatenate([n, rng.uniform(-40, 5, size=(5, 1)).astype(np.float32)], 1)
        a, b = eng.clip_polyhedron(sc["mesh"], pl), oracle.clip(sc["mesh"], pl)
        assert np.array_equal(a["off"], b["off"]) and np.array_equal(a["nbr"], b["nbr"])
        assert np.allclose(a["pos"], b["pos"], rtol=RTOL, atol=1e-5)
    eng.close()


def test_in_plane_cells_and_islands(gpu_engine, oracle):
    from surtr_amd import meshgen
    v, t = meshgen.cube(1.0)
    v2 = np.concatenate([v, v + np.float32([5, 0, 0])])
    t2 = np.concatenate([t, t + 8])
    sc = scenes.make_scene(v2, t2, 6)
    c, got, ref = run_event(gpu_engine, oracle, sc, 3)
    assert got["frag_ids"][:, 2].max() >= 1
    assert_event_equal(got, ref)


def test_repeatable(gpu_engine):
    sc = scenes.blob_scene(64)
    eng = gpu_engine.Engine(0)
    eng.upload_pieces([sc["mesh"]], [sc["convex"]])
    eng.upload_pattern(sc["face_off"], sc["v012"])
    eng.place_cells(sc["scale"], sc["translate"])
    eng.fracture_event(0, 64)
    a = eng.download()
    for _ in range(3):
        eng.fracture_event(0, 64)
        b = eng.download()
        for k in a:
            assert np.array_equal(a[k], b[k]), k
    eng.close()


def test_cpp_host_layer_and_harness(gpu_engine, oracle):
    """The C++ host layer (reference API names over the C ABI) through the headless harness binary."""
    import json as js
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "surtr_amd", "host")])
    exe = os.path.join(root, "surtr_amd", "host", "surtr_harness")
    out = js.loads(subprocess.check_output([exe, "--mesh", "cube", "--cells", "8"]).decode().strip().splitlines()[-1])
    sc = scenes.cube_scene(8)
    planes = oracle.place_cells(sc["v012"], sc["scale"], sc["translate"])
    ref = oracle.event([sc["mesh"]], [sc["convex"]], sc["face_off"], planes)
    assert out["fragments"] == ref["frag_ids"].shape[0] and out["mesh_verts"] == ref["mesh_pos"].shape[0]
    assert out["indices"] == ref["idx"].shape[0] and out["conv_verts"] == ref["conv_pos"].shape[0]
    out = js.loads(subprocess.check_output([exe, "--mesh", "torus", "--cells", "256", "--nu", "100", "--nv", "60"]).decode().strip().splitlines()[-1])
    from surtr_amd import meshgen
    sc = scenes.make_scene(*meshgen.bumpy_torus(100, 60), 256)
    planes = oracle.place_cells(sc["v012"], sc["scale"], sc["translate"])
    ref = oracle.event([sc["mesh"]], [sc["convex"]], sc["face_off"], planes, threads=8)
    assert out["fragments"] == ref["frag_ids"].shape[0] and out["mesh_verts"] == ref["mesh_pos"].shape[0]
    assert out["indices"] == ref["idx"].shape[0] and out["mesh_nbrs"] == ref["mesh_nbr"].shape[0]
    # rows f2 + f3: a mesh file in (LoadModelData conventions), PrepareFracture end to end (--ach), fragments out as OBJ
    import tempfile
    v, t = meshgen.blob(3, scale=2.0)
    with tempfile.TemporaryDirectory() as d:
        src, dst = os.path.join(d, "in.obj"), os.path.join(d, "out.obj")
        with open(src, "w") as f:
            for a in v:
                f.write("v %.9g %.9g %.9g\n" % (-a[0], a[1], a[2]))
            for a, b, c in t:
                f.write("f %d %d %d\n" % (c + 1, b + 1, a + 1))
        out = js.loads(subprocess.check_output([exe, "--obj-in", src, "--cells", "16", "--ach", "--obj", dst]).decode().strip().splitlines()[-1])
        rv, rt = gpu_engine.read_obj(src)
        eng = gpu_engine.Engine(0)
        sc, c, got = scenes.prepare_fracture(eng, rv, rt, n_cells=16)
        eng.close()
        assert out["fragments"] == c.n_frag and out["mesh_verts"] == c.mesh_verts and out["indices"] == c.n_idx and out["conv_verts"] == c.conv_verts
        text = open(dst).read()
        assert text.count("\no ") + 1 == c.n_frag and text.count("\nf ") == c.n_idx // 3


def test_torus_with_ach_convex(gpu_engine, oracle):
    """cfg4 with the reference's initial Convex (ACH: ICH(20) normals -> k-DOP -> clipped 2x box) instead of the plain box."""
    sc = scenes.torus_scene(4096)
    eng = gpu_engine.Engine(0)
    sc["convex"], planes = scenes.ach_convex(eng, sc["mesh"]["pos"])
    eng.close()
    n = oracle.hull_normals(sc["mesh"]["pos"], 20)
    assert planes.shape[0] == 2 * n.shape[0] and sc["convex"]["pos"].shape[0] > 8
    c, got, ref = run_event(gpu_engine, oracle, sc, 3, threads=16)
    assert c.status == 0
    assert_event_equal(got, ref)


@pytest.mark.gpu
def test_prepare_fracture_end_to_end_gpu(gpu_engine, oracle):
    """Row f2 on the GPU: PrepareFracture (Src/Surtr.cpp:1747-1827) of a 2 562-vertex blob into 64 cells."""
    v, t = meshgen.blob(4, scale=70.0)
    eng = gpu_engine.Engine(0)
    sc, c, got = scenes.prepare_fracture(eng, v, t, n_cells=64)
    lo, hi = v.min(0), v.max(0)
    nrm = oracle.hull_normals(v, 20)
    pl = oracle.kdop_planes(v, nrm, ach=True, max_axis_scale=float(max(float(hi[a]) - float(lo[a]) for a in range(3))), gap_inv=2000.0)
    ach = oracle.clip(scenes.box_solid(sc["scale"], sc["translate"]), pl)
    planes = oracle.place_cells(sc["v012"], sc["scale"], sc["translate"])
    ref = oracle.event([oracle.neighbours_from_mesh(v, t)], [ach], sc["face_off"], planes, refit=True, render=True, threads=8)
    assert c.status == 0 and c.n_frag >= 48
    assert_event_equal(got, ref)
    eng.close()


@pytest.mark.parametrize("name", ["pinched_face_fragment", "degenerate_sliver_convex"])
def test_degenerate_fragments_found_by_the_refracture_fuzz(gpu_engine, oracle, name):
    """tests/test_emul_parity.py explains the two fixtures; same checks on the GPU."""
    d = np.load(os.path.join(HERE, name + ".npz"))
    mesh = {"pos": d["mesh_pos"], "off": d["mesh_off"], "nbr": d["mesh_nbr"]}
    conv = {"pos": d["conv_pos"], "off": d["conv_off"], "nbr": d["conv_nbr"]}
    fo = d["fo"].astype(np.uint32) if "fo" in d.files else np.uint32([0, d["planes"].shape[0]])
    eng = gpu_engine.Engine(0)
    eng.upload_pieces([mesh], [conv]); eng.upload_planes(fo, d["planes"])
    c = eng.fracture_event(0, len(fo) - 1, flags=3)
    got = eng.download()
    eng.close()
    ref = oracle.event([mesh], [conv], fo, d["planes"], refit=True, render=True, threads=2)
    assert c.status == 0
    assert_event_equal(got, ref)


# ---- the per-Piece operators, device-resident pieces (tests/test_solid_ops.py holds the cases) ------------------------
import test_solid_ops as _ops
import test_regroup as _rg


@pytest.mark.parametrize("case", _ops.CASES, ids=lambda f: f.__name__)
def test_solid_ops_gpu(gpu_engine, oracle, case):
    case(gpu_engine, oracle)


def test_regroup_then_refit_on_a_two_level_torus_fracture(gpu_engine, oracle):
    """Rows A13 / f1 on the GPU tier: cfg4's torus x 256 cells, its 234 fragments hit by a 32-cell pattern; bind sets,
    HandleConvexIsland on the un-refitted Convex solids, then surtr_event_refit (Src/Surtr.cpp:1921-1939)."""
    _rg.check_regroup_and_refit_order(gpu_engine, oracle, 256, 32, torus=True)


def test_partial_fracture_of_the_torus_pieces(gpu_engine, oracle):
    _rg.check_partial_fracture_merges_out_of_impact(gpu_engine, oracle, 256, 32, torus=True)


def test_outside_mask_skips_pieces_gpu(gpu_engine, oracle):
    sc = scenes.cube_scene(8)
    eng = gpu_engine.Engine(0)
    eng.upload_pieces([sc["mesh"], sc["mesh"]], [sc["convex"], sc["convex"]])
    eng.upload_pattern(sc["face_off"], sc["v012"])
    eng.place_cells(sc["scale"], sc["translate"])
    eng.fracture_event(0, 8, outside=[1, 0], flags=3)
    got = eng.download()
    eng.close()
    planes = oracle.place_cells(sc["v012"], sc["scale"], sc["translate"])
    ref = oracle.event([sc["mesh"]] * 2, [sc["convex"]] * 2, sc["face_off"], planes, outside=[1, 0])
    assert np.all(got["frag_ids"][:, 1] == 1)
    assert_event_equal(got, ref)


def test_errors_are_reported_on_the_gpu(gpu_engine):
    sc = scenes.blob_scene(64)
    # bad topology: refused by the device-side link check of the upload
    bad = dict(sc["mesh"], nbr=sc["mesh"]["nbr"].copy())
    bad["nbr"][0] = (bad["nbr"][0] + 3) % 8
    eng = gpu_engine.Engine(0)
    with pytest.raises(gpu_engine.SurtrError) as e:
        eng.upload_pieces([bad], [sc["convex"]])
    assert e.value.code == gpu_engine.E_TOPOLOGY
    with pytest.raises(gpu_engine.SurtrError) as e:
        eng.fracture_event(0, 1)
    assert e.value.code == gpu_engine.E_STATE
    # arena too small: SURTR_E_CAPACITY, and the context keeps working once the arena is back
    eng.set_arena(64, 256, 256)
    eng.upload_pieces([sc["mesh"]], [sc["convex"]])
    eng.upload_pattern(sc["face_off"], sc["v012"])
    eng.place_cells(sc["scale"], sc["translate"])
    with pytest.raises(gpu_engine.SurtrError) as e:
        eng.fracture_event(0, 64)
    assert e.value.code == gpu_engine.E_CAPACITY
    eng.set_arena(0, 0, 0)
    c = eng.fracture_event(0, 64)
    assert c.status == 0 and c.n_frag == 48
    # a blob buffer that is too small: refused on the host when the counts are known, flagged in the header otherwise
    import torch
    small = torch.zeros(4096, dtype=torch.uint8, device="cuda:0")
    with pytest.raises(gpu_engine.SurtrError) as e:
        eng.pack_dev(small.data_ptr(), 4096)
    assert e.value.code == gpu_engine.E_CAPACITY
    eng.fracture_event_async(0, 64)
    eng.pack_dev(small.data_ptr(), 4096)
    torch.cuda.synchronize()
    hdr = np.frombuffer(small.cpu().numpy()[:36].tobytes(), np.uint32)
    assert hdr[0] == 0 and hdr[7] == gpu_engine.E_CAPACITY
    with pytest.raises(gpu_engine.SurtrError) as e:
        eng.event_counts()
    assert e.value.code == gpu_engine.E_CAPACITY
    eng.close()


def test_image_arena_exhaustion_on_the_torus(gpu_engine, oracle, torus_run, monkeypatch):
    """k_prep_pairs runs out of image room after a few dozen pairs: the rest is pre-passed by k_clip_pairs itself."""
    monkeypatch.setenv("SURTR_IMG_BYTES", str(8 << 20))
    sc, c0, got0, ref = torus_run
    eng = gpu_engine.Engine(0)
    eng.upload_pieces([sc["mesh"]], [sc["convex"]]); eng.upload_pattern(sc["face_off"], sc["v012"]); eng.place_cells(sc["scale"], sc["translate"])
    c = eng.fracture_event(0, 1024, flags=3)
    got = eng.download()
    qs = eng.queue_stats()
    eng.close()
    assert c.status == 0 and qs[16 + 13] > 0, "no pair was left without an image"
    n = c.n_frag
    assert np.array_equal(got["frag_ids"], ref["frag_ids"][:n]) and np.array_equal(got["mesh_nbr"], ref["mesh_nbr"][:c.mesh_nbrs])
    assert np.array_equal(got["idx"], ref["idx"][:c.n_idx])


def test_torus_4096_repeat_run_digest(gpu_engine, torus_run):
    """The arena is filled in arrival order (atomics): three more runs of the whole cfg4 event give the same blob."""
    sc, c0, got0, ref = torus_run
    want = json.load(open(os.path.join(HERE, "digests.json")))["torus4096"]
    eng = gpu_engine.Engine(0)
    eng.upload_pieces([sc["mesh"]], [sc["convex"]]); eng.upload_pattern(sc["face_off"], sc["v012"]); eng.place_cells(sc["scale"], sc["translate"])
    for _ in range(3):
        eng.fracture_event(0, 4096, flags=3)
        got = eng.download()
        for k in TOPO:
            assert hashlib.sha256(np.ascontiguousarray(got[k]).tobytes()).hexdigest() == want[k], k
        assert np.array_equal(got["mesh_pos"], got0["mesh_pos"])
    eng.close()


def test_cpp_api_surface(gpu_engine, oracle):
    """The reference-named C++ API (Poly::ExtractFaces / RenderPolyhedron / Transform / Moments / ClipPolyhedron(polygon3D),
    Kdop::KdopContainer, GenerateICHNormal) through the harness, every result held against the oracle."""
    import json as js
    import subprocess
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "surtr_amd", "host")])
    exe = os.path.join(root, "surtr_amd", "host", "surtr_harness")
    with tempfile.TemporaryDirectory() as d:
        dump = os.path.join(d, "api.json")
        subprocess.check_output([exe, "--mesh", "torus", "--cells", "64", "--nu", "60", "--nv", "36", "--api-dump", dump])
        data = js.load(open(dump))

    def solid(j):
        return {"pos": np.array(j["pos"], np.float32).reshape(-1, 3), "off": np.array(j["off"], np.uint32), "nbr": np.array(j["nbr"], np.int32)}

    def same(a, b):
        assert np.array_equal(a["off"], b["off"]) and np.array_equal(a["nbr"], b["nbr"])
        assert np.allclose(a["pos"], b["pos"], rtol=RTOL, atol=1e-6)
    assert len(data["cases"]) == 4
    for case in data["cases"]:
        mesh, conv = solid(case["mesh"]), solid(case["convex"])
        fo, fi = oracle.extract_faces(mesh)
        assert [fi[fo[i]:fo[i + 1]].tolist() for i in range(len(fo) - 1)] == case["faces"]
        for key, s, cv in (("render_mesh", mesh, False), ("render_convex", conv, True)):
            rv, ri = oracle.render(s, convex=cv, colour=(0.5, 0.25, 1.0))
            assert case[key]["nv"] == s["pos"].shape[0] + 1                       # appended after the vertex already there
            assert (np.array(case[key]["idx"], np.int64) - 1).tolist() == ri.tolist()
            assert case[key]["color"] == [0.5, 0.25, 1.0]
        ref = oracle.refit(conv, mesh, 4)
        same(solid(case["refit_task"]), ref)
        same(solid(case["refit_kdop"]), ref)                                       # the task spelled with Kdop::KdopContainer
        vol, cen = oracle.moments(mesh)
        assert abs(case["volume"] - vol) <= 1e-9 * abs(vol) and np.allclose(case["centroid"], cen, rtol=1e-6, atol=1e-7)
        same(solid(case["moved"]), oracle.transform(mesh, np.array(case["world"], np.float32).reshape(4, 4)))
    for cc in data["cell_clip"]:
        same(solid(cc["by_polygon"]), solid(cc["by_planes"]))
        same(solid(cc["by_polygon"]), oracle.clip(solid(cc["box"]), np.array(cc["planes"], np.float32).reshape(-1, 4)))
        assert solid(cc["by_polygon"])["pos"].shape[0] >= 4
    # Surtr::DoFracture with PartialFracture (Src/Surtr.cpp:1885-1959) on the device path: the bind sets against the oracle's
    # ApplyFracture / MergeOutOfImpact / HandleConvexIsland on the same un-refitted Convex solids, the pieces that came out
    # against the oracle's refit; MergeOutOfImpact + HandleConvexIsland called on their own give the same sets
    df = data["do_fracture"]
    conv = [solid(c["s"]) for c in df["convex"]]
    cell = np.array(df["piece_cell"], np.int32)
    n_out = df["n_outside"]
    assert 0 < n_out < df["n_pieces_in"] and len(conv) > n_out + 10
    cloud = np.array(df["cloud"], np.float32).reshape(-1, 3)
    ro, rp = oracle.regroup(conv, cell, n_outside=n_out, partial=True, sphere_points=cloud, origin=np.array(df["origin"], np.float32), radius=df["radius"])
    want = [sorted(rp[ro[c]:ro[c + 1]].tolist()) for c in range(len(ro) - 1)]
    assert [sorted(b) for b in df["binds"]] == want
    assert [sorted(b) for b in df["binds_by_set_functions"]] == want
    assert len(want[0]) >= n_out and max(len(b) for b in want[1:]) > 1
    out = df["pieces_out"]
    flat = [p for b in want for p in b]                      # compounds in order, pieces ascending inside: the order of pieces_out
    assert len(out) == len(flat) == len(conv)
    for k, p in enumerate(flat):
        got_c = solid(out[k]["convex"])
        if p < n_out:
            same(got_c, conv[p])                                 # a piece the event skipped comes back as it was
        elif k % 7 == 0:
            same(got_c, oracle.refit(conv[p], solid(out[k]["mesh"]), 4))
    # the three tasks through the reference's signatures (Inc/Surtr.h:270-272): one placed cell, piece 1 outside
    tk = data["do_fracture"]["tasks"]
    planes = np.array(tk["cell_planes"], np.float32).reshape(-1, 4)
    meshes = [solid(t["mesh"]) for t in tk["target"]]; convs = [solid(t["convex"]) for t in tk["target"]]
    om = np.zeros(len(meshes), np.uint8); om[1] = 1
    ref = oracle.event(meshes, convs, np.array([0, planes.shape[0]], np.uint32), planes, outside=om, refit=False, render=False)
    assert len(tk["fractured"]) == ref["frag_ids"].shape[0] > 0
    ref2 = oracle.event(meshes, convs, np.array([0, planes.shape[0]], np.uint32), planes, outside=om, refit=True, render=True)
    for k, fr in enumerate(tk["fractured"]):
        same(solid(fr["mesh"]), fragment(ref, k, "mesh")); same(solid(fr["convex"]), fragment(ref, k, "conv"))
        same(solid(tk["refitted"][k]["convex"]), fragment(ref2, k, "conv"))
        a, b = int(ref2["idx_off"][k]), int(ref2["idx_off"][k + 1])
        assert tk["init"][k]["idx"] == ref2["idx"][a:b].tolist() and tk["init"][k]["nv"] == fragment(ref2, k, "mesh")["pos"].shape[0]
        assert tk["init"][k]["points"] == fragment(ref2, k, "conv")["pos"].shape[0]


# ---- Voronoi cells on the device (row A2) -------------------------------------------------------------------------------
import test_build_cells as _bc


@pytest.mark.parametrize("n,groups", [(8, 1), (64, 1), (1024, 1), (4096, 1), (32, 234)])
def test_build_cells_gpu(gpu_engine, oracle, n, groups):
    """Row A2 on the device against the oracle's cells (structure equal, coordinates to 1e-12) and the host builder (bit for bit)."""
    _bc.check_cells(gpu_engine, n, groups, oracle)


def test_build_cells_time_and_event(gpu_engine, torus_run):
    """4 096 cells in a few milliseconds (the host builder takes seconds), and the cfg4 event on them gives the golden digests."""
    import time
    sc, c0, got0, ref = torus_run
    eng = gpu_engine.Engine(0)
    eng.build_cells(sc["seeds"])
    t0 = time.perf_counter()
    for _ in range(3):
        eng.build_cells(sc["seeds"])
    ms = (time.perf_counter() - t0) / 3 * 1e3
    print("surtr_build_cells, 4096 cells: %.2f ms per call (host wall, incl. the size read-back)" % ms)
    assert ms < 50.0
    eng.upload_pieces([sc["mesh"]], [sc["convex"]])
    eng.place_cells(sc["scale"], sc["translate"])
    eng.fracture_event(0, 4096, flags=3)
    got = eng.download()
    eng.close()
    for k in TOPO:
        assert np.array_equal(got[k], got0[k]), k


def test_one_bad_fragment_does_not_fail_the_event(gpu_engine, oracle):
    import test_emul_parity as _ep
    _ep.check_nonterminating_faces_fragment(gpu_engine, oracle)


def test_refracture_fuzz_case_264_whole(gpu_engine, oracle):
    """The whole event the fragment comes from (200 first-level cells of a 235 x 196 torus, 8 cells per piece): SURTR_OK,
    one flagged fragment, everything else bit-equal."""
    from test_refracture import _refracture
    c, got, ref, npieces = _refracture(gpu_engine, oracle, 200, 8, 235, 196)
    assert c.status == 0 and c.n_failed == 1 and int((got["frag_status"] != 0).sum()) == 1
    assert_event_equal(got, ref)


def test_degenerate_walk_bound_pair(gpu_engine, oracle):
    """Refracture fuzz seed 555, case 110: a sliver piece whose Convex the reference clips to nothing through a relink walk
    that hits its step bound (tests/test_literal_clip.py) -- single solids, then the whole event (200 first-level cells of a
    189 x 91 torus, 3 cells per piece)."""
    import test_literal_clip as _lc
    from test_refracture import _refracture
    _lc.check_degenerate_pair(gpu_engine, oracle)
    c, got, ref, npieces = _refracture(gpu_engine, oracle, 200, 3, 189, 91)
    assert c.status == 0 and c.n_failed == 0
    assert_event_equal(got, ref)


def test_sliver_convex_walk_bound(gpu_engine, oracle):
    """Refracture fuzz seed 13579, case 84 (tests/test_literal_clip.py): the single Convex, then the whole event (200 first-level
    cells of a 211 x 107 torus, 32 cells per piece: 4 318 fragments, not 4 319)."""
    import test_literal_clip as _lc
    from helpers import assert_event_equal_flagged
    from test_refracture import _refracture
    _lc.check_sliver_convex_walk_bound(gpu_engine, oracle)
    c, got, ref, npieces = _refracture(gpu_engine, oracle, 200, 32, 211, 107)
    assert c.status == 0
    assert_event_equal_flagged(got, ref)


def test_stale_id_sliver_mesh(gpu_engine, oracle):
    """Refracture fuzz seed 90210, case 1271 (tests/test_literal_clip.py): the single pair, then the whole event."""
    import test_literal_clip as _lc
    from helpers import assert_event_equal_flagged
    from test_refracture import _refracture
    _lc.check_stale_id_pair(gpu_engine, oracle)
    c, got, ref, npieces = _refracture(gpu_engine, oracle, 96, 3, 54, 150)
    assert c.status == 0
    assert_event_equal_flagged(got, ref)


def test_refit_result_that_is_no_polyhedron(gpu_engine, oracle):
    """Refracture fuzz seed 555002, case 82: one fragment's refit ends, in the reference, as a Convex with a one-way link;
    the single fragment first (tests/test_emul_parity.py), then the whole event (200 first-level cells of a 53 x 25 torus,
    17 cells per piece): SURTR_OK, that fragment flagged (it keeps its un-refitted Convex: the degenerate policy), every other
    fragment equal to the restated reference."""
    import test_emul_parity as _ep
    from helpers import assert_event_equal_flagged
    from test_refracture import _refracture
    _ep.check_refit_invalid_in_reference(gpu_engine, oracle)
    c, got, ref, npieces = _refracture(gpu_engine, oracle, 200, 17, 53, 25)
    assert c.status == 0 and c.n_failed == 1 and int(np.count_nonzero(got["frag_status"])) == 1 and not got["flagged_pairs"]
    assert_event_equal_flagged(got, ref)


@pytest.mark.parametrize("name,n", [("urchin64", 64), ("urchin1024", 1024)])
def test_deep_lobed_mesh_islands(gpu_engine, oracle, name, n):
    """cfg2 / cfg3 cell counts on the deep-lobed mesh (meshgen.urchin): at least a tenth of the non-empty cells hold two or
    more islands, faces are far from convex; full equality with the oracle and the committed digests."""
    c, got, ref = run_event(gpu_engine, oracle, scenes.urchin_scene(n), 3, threads=8)
    assert c.status == 0 and c.n_failed == 0
    assert_event_equal(got, ref)
    ids = got["frag_ids"]
    cells = np.unique(ids[:, 0])
    multi = sum(1 for cc in cells if int((ids[:, 0] == cc).sum()) > 1)
    assert multi >= 0.1 * cells.shape[0], (multi, cells.shape[0])
    want = json.load(open(os.path.join(HERE, "digests.json")))[name]
    assert c.n_frag == want["n_frag"] and c.mesh_verts == want["mesh_verts"] and c.n_idx == want["n_idx"]
    for k in TOPO:
        assert hashlib.sha256(np.ascontiguousarray(got[k]).tobytes()).hexdigest() == want[k], k


def test_device_rings_gpu(gpu_engine, oracle):
    """Row f3: ExtractNeighborFromMesh (Src/Poly.cpp:128-263) on the device against the oracle's rings; the 100 000-triangle torus
    in a few milliseconds."""
    import test_host_helpers as _hh
    _hh.check_device_rings(gpu_engine, oracle)
    v, t = meshgen.bumpy_torus()
    want = gpu_engine.neighbors_from_mesh(v, t)
    ora = oracle.neighbours_from_mesh(v, t)
    assert np.array_equal(want["off"], ora["off"]) and np.array_equal(want["nbr"], ora["nbr"])
    eng = gpu_engine.Engine(0)
    eng.neighbors_from_mesh(v, t)
    got, ms = eng.neighbors_from_mesh(v, t)
    eng.close()
    assert np.array_equal(got["off"], want["off"]) and np.array_equal(got["nbr"], want["nbr"])
    print("surtr_neighbors_from_mesh_dev, 100 000 triangles: %.3f ms of kernels" % ms)
    assert 0 < ms < 5.0


def test_cpp_rccl_harness_single_rank(oracle):
    """The C++ exchange step (surtr_rccl.cpp: ncclAllGather of sizes, then of the packed blobs) on a 1-rank communicator:
    what one GPU can rehearse; the N-rank run forks one process per GPU (surtr_harness_mgpu --ranks N)."""
    import json as js
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "surtr_amd", "host")])
    exe = os.path.join(root, "surtr_amd", "host", "surtr_harness_mgpu")
    out = js.loads(subprocess.check_output([exe, "--ranks", "1", "--cells", "256", "--nu", "100", "--nv", "60", "--steps", "2"], timeout=300).decode().strip().splitlines()[-1])
    sc = scenes.make_scene(*meshgen.bumpy_torus(100, 60), 256)
    v = sc["mesh"]["pos"]; lo, hi = v.min(0), v.max(0)
    nrm = oracle.hull_normals(v, 20)
    pl = oracle.kdop_planes(v, nrm, ach=True, max_axis_scale=float(max(float(hi[a]) - float(lo[a]) for a in range(3))), gap_inv=2000.0)
    ach = oracle.clip(scenes.box_solid(sc["scale"], sc["translate"]), pl)
    planes = oracle.place_cells(sc["v012"], sc["scale"], sc["translate"])
    ref = oracle.event([sc["mesh"]], [ach], sc["face_off"], planes, threads=8)
    assert out["ranks"] == 1 and out["fragments"] == ref["frag_ids"].shape[0] and out["per_rank"] == [ref["frag_ids"].shape[0]]


def test_wide_global_scratch_variant_gpu(gpu_engine, oracle):
    """A band that outgrows even the double-size LDS topology (a 350 000-vertex torus cut by 8 large cells): the pair is clipped
    by the same templated code on global scratch with 32-bit indices (Topo<InGlobal>).  Round 1 had this on the emulation only."""
    v, t = meshgen.bumpy_torus(700, 500)
    sc = scenes.make_scene(v, t, 8)
    eng = gpu_engine.Engine(0)
    eng.upload_pieces([sc["mesh"]], [sc["convex"]]); eng.upload_pattern(sc["face_off"], sc["v012"]); eng.place_cells(sc["scale"], sc["translate"])
    c = eng.fracture_event(0, 8, flags=3)
    got = eng.download()
    qs = eng.queue_stats()
    eng.close()
    assert c.status == 0 and qs[16 + 15] > 0, "no pair took the global-scratch variant: %s" % qs[16:32].tolist()
    planes = oracle.place_cells(sc["v012"], sc["scale"], sc["translate"])
    ref = oracle.event([sc["mesh"]], [sc["convex"]], sc["face_off"], planes, threads=8)
    assert_event_equal(got, ref)


def test_fracture_pattern_around_an_impact_point(gpu_engine, oracle):
    """Surtr::GenerateFracturePattern + DoFracture's placement (Src/Surtr.cpp:2072-2096, 1887-1896): cells dense around the
    pattern origin (exponential seed distances), the pattern scaled by 2 x MaxAxisScale and moved to an impact point on the
    surface; cells built on the device."""
    sc = scenes.blob_scene(8)
    seeds = scenes.pattern_seeds(128, 0.05)
    v = sc["mesh"]["pos"]
    lo, hi = v.min(0), v.max(0)
    max_axis = np.float32(max(float(hi[a]) - float(lo[a]) for a in range(3)))
    scale = np.full(3, np.float32(2.0) * max_axis, np.float32)
    impact = v[np.argmax(v[:, 0])].astype(np.float32)
    eng = gpu_engine.Engine(0)
    eng.upload_pieces([sc["mesh"]], [sc["convex"]])
    eng.build_cells(seeds)
    cells = eng.download_cells()
    ref_cells = gpu_engine.voronoi_cells(seeds)
    assert np.array_equal(cells["face_gen"], ref_cells["face_gen"]) and np.array_equal(cells["verts"], ref_cells["verts"].reshape(-1, 3))
    eng.place_cells(scale, impact)
    c = eng.fracture_event(0, 128, flags=3)
    got = eng.download()
    eng.close()
    planes = oracle.place_cells(cells["v012"], scale, impact)
    ref = oracle.event([sc["mesh"]], [sc["convex"]], cells["cell_face_off"], planes, threads=8)
    assert c.status == 0 and c.n_frag > 20
    assert_event_equal(got, ref)


import test_edge_cases as _ec


@pytest.mark.parametrize("case", _ec.CASES, ids=lambda f: f.__name__)
def test_edge_cases_gpu(gpu_engine, oracle, case):
    import inspect
    case(gpu_engine, oracle) if len(inspect.signature(case).parameters) == 2 else case(gpu_engine)


def test_graft_entry_smoke():
    """__graft_entry__.smoke(): the one small event the driver runs before the bench."""
    import __graft_entry__
    __graft_entry__.smoke()


@pytest.mark.gpu
@pytest.mark.parametrize("world,in_flight", [(8, 1), (4, 1), (2, 1), (8, 6), (4, 6)])
def test_sharded_torus_event_equals_the_whole(gpu_engine, oracle, torus_run, world, in_flight):
    """The configs[3] event cut into `world` contiguous cell blocks, as the ranks of a strong-sharded run take them (one
    after the other on this GPU): blocks of at most 2 048 pairs go through k_prep_pairs_wide -- or, when the context has been
    told that several of them share the GPU (surtr_set_events_in_flight, as bench.py's are), through the regular pre-pass beside
    k_clip_convex and the record clipper + catcher.  Merged in rank order they are the whole event, bit for bit."""
    sc, c_all, got_all, ref = torus_run
    eng = gpu_engine.Engine(0)
    try:
        eng.set_events_in_flight(in_flight)
        eng.upload_pieces([sc["mesh"]], [sc["convex"]])
        eng.upload_pattern(sc["face_off"], sc["v012"])
        eng.place_cells(sc["scale"], sc["translate"])
        parts = []
        for r in range(world):
            b, e = gpu_engine.cell_block(r, world, sc["n_cells"])
            c = eng.fracture_event(b, e, flags=3)
            assert c.status == 0
            parts.append(eng.download())
        merged = gpu_engine.merge_fragments(parts)
    finally:
        eng.close()
    assert merged["frag_ids"].shape[0] == got_all["frag_ids"].shape[0]
    for k in merged:
        assert np.array_equal(np.asarray(merged[k]), np.asarray(got_all[k])), k
