"""The pre-pass of k_prep_pairs on the sorted copy (surtr_amd/csrc/prep_sorted.h, round 4): sphere hierarchy, first clipping planes
by look-up, record images for the record clipper, the small tier.  Whatever selects the band and however it leaves the kernel,
the event must be the oracle's -- Poly::ClipPolyhedron's order rules (Src/Poly.cpp:333-357, 464-495) bit for bit.
CPU tier: the single-lane emulation (its thresholds send meshes of 48 vertices and more through k_prep_pairs); GPU tier:
BASELINE configs[3] against the committed digests with every combination of the round's switches."""
import hashlib
import json
import os

import numpy as np
import pytest

from helpers import assert_event_equal
from surtr_amd import meshgen, scenes
from test_record_clipper import _event, TOPO, HERE


def _quad_torus(nu=40, nv=24, R=1.0, r=0.35):
    """A torus of QUADS as a solid (rings of four neighbours, faces that are no triangles): the pre-pass must walk the faces."""
    u = np.arange(nu) * (2 * np.pi / nu); v = np.arange(nv) * (2 * np.pi / nv)
    uu, vv = np.meshgrid(u, v, indexing="ij")
    pos = np.stack([(R + r * np.cos(vv)) * np.cos(uu), (R + r * np.cos(vv)) * np.sin(uu), r * np.sin(vv)], -1).reshape(-1, 3).astype(np.float32)
    idx = lambda i, j: (i % nu) * nv + (j % nv)
    nbr = np.zeros((nu * nv, 4), np.int32)
    for i in range(nu):
        for j in range(nv):
            # counter-clockwise seen from outside (the winding of meshgen.bumpy_torus's triangles, checked by the volume below)
            nbr[idx(i, j)] = [idx(i + 1, j), idx(i, j + 1), idx(i - 1, j), idx(i, j - 1)]
    return {"pos": pos, "off": (np.arange(nu * nv + 1) * 4).astype(np.uint32), "nbr": nbr.reshape(-1)}


@pytest.mark.parametrize("env", [{}, {"SURTR_REC_MAXN": "100000"}, {"SURTR_REC": "0"}, {"SURTR_PREP_SORTED": "0"},
                                 {"SURTR_SMALL": "1", "SURTR_REC_MAXN": "100000"},
                                 # (an event of 256 pairs clips the Convexes and prepares the bands side by side: here one after the other)
                                 {"SURTR_FRONT_PAR": "0"}])
def test_sorted_prepass_emulation_equals_oracle(emul_engine, oracle, monkeypatch, env):
    monkeypatch.setenv("SURTR_WAVE", "1")
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    sc = scenes.make_scene(*meshgen.bumpy_torus(100, 60), 256)
    c, got, ref, qs = _event(emul_engine, oracle, sc, 96)
    assert c.status == 0 and c.n_frag == ref["frag_ids"].shape[0] > 30
    assert_event_equal(got, ref)
    assert np.array_equal(got["mesh_pos"], ref["mesh_pos"])
    sorted_pairs, rec_images = int(qs[92]), int(qs[91])
    if env.get("SURTR_PREP_SORTED") == "0":
        assert sorted_pairs == 0 and rec_images == 0
    else:
        assert sorted_pairs > 50
        assert (rec_images == 0) if env.get("SURTR_REC") == "0" else (rec_images > 20)


def test_events_in_flight_hint_changes_the_kernels_not_the_event(emul_engine, oracle):
    """surtr_set_events_in_flight(4): an event of 96 pairs takes the record clipper + catcher instead of the general clipper
    (it would not by its size); the result is the oracle's either way."""
    sc = scenes.make_scene(*meshgen.bumpy_torus(100, 60), 256)
    eng = emul_engine.Engine(0)
    try:
        eng.upload_pieces([sc["mesh"]], [sc["convex"]]); eng.upload_pattern(sc["face_off"], sc["v012"]); eng.place_cells(sc["scale"], sc["translate"])
        c1 = eng.fracture_event(0, 96, flags=3); q1 = eng.queue_stats(); g1 = eng.download()
        eng.set_events_in_flight(4)
        c2 = eng.fracture_event(0, 96, flags=3); q2 = eng.queue_stats(); g2 = eng.download()
    finally:
        eng.close()
    planes = oracle.place_cells(sc["v012"], sc["scale"], sc["translate"])
    ref = oracle.event([sc["mesh"]], [sc["convex"]], sc["face_off"], planes, cell_end=96)
    assert c1.status == 0 and c2.status == 0
    assert_event_equal(g1, ref); assert_event_equal(g2, ref)
    assert int(q2[88]) > 30, "with the hint the record clipper takes the pairs"


@pytest.mark.parametrize("lib", ["libsurtr_emul.so", "libsurtr_emul_rec.so"])
def test_small_tier_hands_pairs_on_and_the_event_stands(emul_lib_path, oracle, monkeypatch, lib):
    """k_clip_pairs_rec (three workgroups per CU, no general clipper inside): with little room its pairs leave their stage in
    global memory, and what it still cannot finish comes back through the large tier from the piece."""
    from surtr_amd import engine
    monkeypatch.setenv("SURTR_WAVE", "1"); monkeypatch.setenv("SURTR_SMALL", "1"); monkeypatch.setenv("SURTR_REC_MAXN", "100000")
    engine._use_library_for_tests(os.path.join(os.path.dirname(emul_lib_path), lib))
    try:
        sc = scenes.make_scene(*meshgen.bumpy_torus(100, 60), 256)
        c, got, ref, qs = _event(engine, oracle, sc, 96)
    finally:
        engine._use_library_for_tests(None)
    assert c.status == 0
    assert_event_equal(got, ref)
    assert int(qs[91]) > 20 and int(qs[88]) > 0


def test_faces_that_are_no_triangles_take_the_face_walks(emul_engine, oracle, monkeypatch):
    monkeypatch.setenv("SURTR_WAVE", "1"); monkeypatch.setenv("SURTR_REC_MAXN", "100000")
    mesh = _quad_torus()
    vol, _ = emul_engine.moments(mesh)
    assert vol > 0.5
    sc = scenes.make_scene(*meshgen.bumpy_torus(40, 24), 24)       # (cells + convex of a torus of the same size)
    sc["mesh"] = mesh
    sc["convex"] = scenes.box_solid(mesh["pos"].max(0) - mesh["pos"].min(0), (mesh["pos"].max(0) + mesh["pos"].min(0)) / 2)
    c, got, ref, qs = _event(emul_engine, oracle, sc, 24)
    assert c.status == 0 and c.n_frag == ref["frag_ids"].shape[0] > 10
    assert_event_equal(got, ref)
    assert int(qs[92]) > 10        # (through the sorted pre-pass, whose look-ups cannot decide these vertices alone)


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{}, {"SURTR_REC_MAXN": "100000"}, {"SURTR_REC": "0"}, {"SURTR_PREP_SORTED": "0"}, {"SURTR_SMALL": "1"},
                                 # (the arrangement of small events -- k_clip_convex beside the pre-pass kernel -- on the large one)
                                 {"SURTR_FRONT_PAR": "1"},
                                 # (what bench.py's contexts run with: six of them on the GPU -- two polling catchers, a quarter of the faces tier)
                                 {"SURTR_EVENTS_IN_FLIGHT": "6"}])
def test_torus_4096_digest_whatever_the_prepass(gpu_engine, monkeypatch, env):
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    want = json.load(open(os.path.join(HERE, "digests.json")))["torus4096"]
    sc = scenes.torus_scene(4096)
    eng = gpu_engine.Engine(0)
    try:
        eng.upload_pieces([sc["mesh"]], [sc["convex"]]); eng.upload_pattern(sc["face_off"], sc["v012"]); eng.place_cells(sc["scale"], sc["translate"])
        c = eng.fracture_event(0, 4096, flags=3)
        qs = eng.queue_stats()
        got = eng.download()
    finally:
        eng.close()
    assert c.status == 0 and c.n_frag == want["n_frag"] and c.mesh_verts == want["mesh_verts"] and c.n_idx == want["n_idx"]
    for k in TOPO:
        assert hashlib.sha256(np.ascontiguousarray(got[k]).tobytes()).hexdigest() == want[k], k
    if env.get("SURTR_PREP_SORTED") != "0":
        assert int(qs[92]) == 4096
        if env.get("SURTR_REC") != "0":
            assert int(qs[91]) > 2000
