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
