"""Product host helpers (C ABI, no GPU) against the oracle and against Qhull."""
import ctypes
import os
import re

import numpy as np
import pytest

from surtr_amd import engine, meshgen, scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_abi_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "surtr_hip.h")).read()
    names = sorted(set(re.findall(r"\b(surtr_[a-z_0-9]+)\s*\(", hdr)))
    assert len(names) >= 18
    assert os.path.exists(engine.lib_path()), "libsurtr_hip.so missing: run __graft_entry__.build()"
    L = ctypes.CDLL(engine.lib_path())
    for n in names:
        assert hasattr(L, n), n


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    engine._use_library_for_tests(None)
    with pytest.raises(engine.SurtrError) as e:
        engine.Engine(0)
    assert e.value.code == engine.E_NOGPU


@pytest.mark.parametrize("name", ["cube", "blob", "torus"])
def test_neighbour_rings_match_oracle(oracle, name):
    v, t = {"cube": meshgen.cube, "blob": meshgen.blob, "torus": lambda: meshgen.bumpy_torus(60, 40)}[name]()
    a = engine.neighbors_from_mesh(v, t)
    b = oracle.neighbours_from_mesh(v, t)
    assert np.array_equal(a["off"], b["off"]) and np.array_equal(a["nbr"], b["nbr"])
    fo, fi = oracle.extract_faces(a)
    assert np.all(np.diff(fo) == 3) and fo.shape[0] - 1 == t.shape[0]       # faces of the solid = the triangles
    assert oracle.moments(a)[0] > 0


def test_neighbour_rings_reject_open_mesh():
    v, t = meshgen.cube()
    with pytest.raises(engine.SurtrError) as e:
        engine.neighbors_from_mesh(v, t[:-1])
    assert e.value.code == engine.E_TOPOLOGY


def test_seed_generators_match_libstdcxx(oracle):
    assert np.array_equal(scenes.uniform_seeds(100), oracle.seeds(100))
    assert np.array_equal(scenes.pattern_seeds(128, 0.01), oracle.seeds(128, mode=1, mean=0.01))
    assert np.array_equal(scenes.pattern_seeds(64, 1.0), oracle.seeds(64, mode=1, mean=1.0))


@pytest.mark.parametrize("n", [8, 64, 200])
def test_voronoi_cells_match_oracle_and_qhull(oracle, n):
    seeds = scenes.uniform_seeds(n)
    a = engine.voronoi_cells(seeds)
    b = oracle.voronoi_cells(seeds)
    for k in ("cell_face_off", "face_gen", "face_vert_off"):
        assert np.array_equal(a[k], b[k]), k
    assert np.abs(a["verts"] - b["verts"]).max() < 1e-12
    # independent geometry check: cell volumes sum to the unit box; every cell vertex is equidistant to its generators
    from scipy.spatial import ConvexHull
    total = 0.0
    for c in range(n):
        f0, f1 = a["cell_face_off"][c], a["cell_face_off"][c + 1]
        pts = a["verts"][a["face_vert_off"][f0]:a["face_vert_off"][f1]]
        total += ConvexHull(pts).volume
        for f in range(f0, f1):
            g = a["face_gen"][f]
            if g < n:
                fv = a["verts"][a["face_vert_off"][f]:a["face_vert_off"][f + 1]]
                d0 = np.linalg.norm(fv - seeds[c], axis=1)
                d1 = np.linalg.norm(fv - seeds[g], axis=1)
                assert np.abs(d0 - d1).max() < 1e-12
    assert abs(total - 1.0) < 1e-9


def test_cell_planes_point_outward(oracle):
    sc = scenes.blob_scene(64)
    planes = oracle.place_cells(sc["v012"], sc["scale"], sc["translate"])
    for c in range(64):
        p = planes[sc["face_off"][c]:sc["face_off"][c + 1]]
        seed = (sc["seeds"][c] * sc["scale"] + sc["translate"]).astype(np.float32)
        assert np.all(p[:, :3] @ seed + p[:, 3] < 0)


def test_moments_matches_oracle_and_partitions(oracle):
    """Row A14: Poly::Moments (Src/Poly.cpp:55-87) on the host = the oracle's; box KAT from SURVEY (volume 0.5)."""
    from surtr_amd import engine, meshgen, scenes
    box = oracle.unit_box()
    v, c = engine.moments(box)
    assert abs(v - 1.0) < 1e-6 and np.allclose(c, 0.0, atol=1e-6)
    half = oracle.clip(box, np.float32([[1, 1, 1, 0]]))
    v, c = engine.moments(half)
    ov, oc = oracle.moments(half)
    assert v == ov and np.array_equal(c, oc.astype(np.float32))
    assert abs(v - 0.5) < 1e-6 and np.allclose(c, -0.135417, atol=1e-5)
    vb, tb = meshgen.blob(2, scale=3.0)
    mesh = engine.neighbors_from_mesh(vb, tb)
    v, c = engine.moments(mesh)
    ov, oc = oracle.moments(mesh)
    assert v == ov and np.array_equal(c, oc.astype(np.float32))


def test_obj_reader_conventions_and_writer(tmp_path):
    """Row f3: OBJ in with the reference's load conventions (x negated, winding flipped, identical positions joined,
    Src/Surtr.cpp:2683-2727) and fragments out as OBJ."""
    from surtr_amd import engine, meshgen
    v, t = meshgen.blob(2, scale=3.0)
    p = tmp_path / "blob.obj"
    with open(p, "w") as f:
        f.write("# test\n")
        for a in v:
            f.write("v %.9g %.9g %.9g\n" % (-a[0], a[1], a[2]))          # pre-mirrored, so that the reader gives v back
        for a, b, c in t:
            f.write("f %d/1/1 %d//2 %d\n" % (c + 1, b + 1, a + 1))       # pre-flipped, corners with /vt/vn decorations
        f.write("v 9 9 9\n")                                            # unused vertex: not emitted
    rv, rt = engine.read_obj(str(p))
    # vertices come back in order of first use by a face, triangles in file order
    order = []
    seen = set()
    for tri in t[:, ::-1]:
        for i in tri:
            if int(i) not in seen:
                seen.add(int(i)); order.append(int(i))
    assert rv.shape[0] == len(order) and rt.shape == t.shape
    assert np.array_equal(rv, v[order])
    inv = np.empty(len(order), np.int64); inv[order] = np.arange(len(order))
    assert np.array_equal(rt, inv[t])
    # the mesh is closed and consistently wound: the neighbour extraction accepts it
    mesh = engine.neighbors_from_mesh(rv, rt)
    assert engine.moments(mesh)[0] > 0
    # scale / translate, negative indices, quads as fans
    q = tmp_path / "quad.obj"
    q.write_text("v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nf -4 -3 -2 -1\n")
    qv, qt = engine.read_obj(str(q), scale=(2, 2, 2), translate=(1, 0, 0))
    assert np.array_equal(qv, np.float32([[1, 0, 0], [-1, 0, 0], [-1, 2, 0], [1, 2, 0]]))
    assert np.array_equal(qt, np.int32([[2, 1, 0], [3, 2, 0]]))
    # writer: one object per fragment, 1-based indices continuing across objects
    fr = {"frag_ids": np.int32([[0, 0, 0], [1, 0, 0]]), "mesh_vert_off": np.uint32([0, 3, 6]),
          "vnc": np.arange(54, dtype=np.float32).reshape(6, 9), "idx_off": np.uint32([0, 3, 6]), "idx": np.uint32([0, 1, 2, 0, 2, 1])}
    w = tmp_path / "out.obj"
    engine.write_obj(str(w), fr)
    lines = w.read_text().splitlines()
    assert lines[0] == "o cell0_piece0_island0" and lines[4] == "f 1 2 3" and lines[5] == "o cell1_piece0_island0" and lines[9] == "f 4 6 5"


def check_device_rings(E, oracle=None):
    """surtr_neighbors_from_mesh_dev == oracle.neighbours_from_mesh == surtr_neighbors_from_mesh (ring order and rotation), and
    the same refusals."""
    from surtr_amd import meshgen
    eng = E.Engine(0)
    try:
        for v, t in (meshgen.cube(), meshgen.blob(2), meshgen.bumpy_torus(40, 24)):
            want = E.neighbors_from_mesh(v, t)
            got, _ = eng.neighbors_from_mesh(v, t)
            assert np.array_equal(got["off"], want["off"]) and np.array_equal(got["nbr"], want["nbr"])
            if oracle is not None:
                o = oracle.neighbours_from_mesh(v, t)
                assert np.array_equal(got["off"], o["off"]) and np.array_equal(got["nbr"], o["nbr"])
        v, t = meshgen.cube()
        bad = t.copy(); bad[0] = bad[0][::-1]                 # one triangle wound the other way: a directed edge twice
        with pytest.raises(E.SurtrError) as e:
            eng.neighbors_from_mesh(v, bad)
        assert e.value.code == E.E_TOPOLOGY
        with pytest.raises(E.SurtrError) as e:
            eng.neighbors_from_mesh(v, t[:-1])                # open surface
        assert e.value.code == E.E_TOPOLOGY
    finally:
        eng.close()


def test_device_rings_emulated(emul_engine, oracle):
    check_device_rings(emul_engine, oracle)
