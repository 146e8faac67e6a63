"""The record clipper (surtr_amd/csrc/wave_clip.h, kernel k_clip_pairs_wave): the regular planes of the Mesh clip on 16-byte
records with the band streamed bucket by bucket.  It must give the general clipper's -- i.e. the oracle's -- event bit for
bit, whichever pairs it takes and whichever it hands to the general clipper in place (Src/Poly.cpp:265-500 either way).
CPU tier: the single-lane emulation with two capacity settings (pairs that fit; pairs that run out of room mid-way, with
pointer jumping from the first walk step); GPU tier: BASELINE configs[3] with the clipper forced on and off."""
import hashlib
import json
import os

import numpy as np
import pytest

from helpers import assert_event_equal, fragment
from surtr_amd import meshgen, scenes

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOPO = ("frag_ids", "mesh_vert_off", "mesh_nbr_off", "mesh_nbr", "conv_vert_off", "conv_nbr_off", "conv_nbr", "idx_off", "idx")


def _event(engine_mod, oracle, sc, cells, flags=3, threads=8):
    eng = engine_mod.Engine(0)
    try:
        eng.upload_pieces([sc["mesh"]], [sc["convex"]])
        eng.upload_pattern(sc["face_off"], sc["v012"])
        eng.place_cells(sc["scale"], sc["translate"])
        c = eng.fracture_event(0, cells, flags=flags)
        qs = eng.queue_stats()
        got = eng.download()
    finally:
        eng.close()
    planes = oracle.place_cells(sc["v012"], sc["scale"], sc["translate"])
    ref = oracle.event([sc["mesh"]], [sc["convex"]], sc["face_off"], planes, refit=bool(flags & 1), render=bool(flags & 2),
                       threads=threads, cell_end=cells)
    return c, got, ref, qs


@pytest.mark.parametrize("lib", ["libsurtr_emul.so", "libsurtr_emul_rec.so"])
def test_record_clipper_emulation_torus(emul_lib_path, oracle, monkeypatch, lib):
    from surtr_amd import engine
    monkeypatch.setenv("SURTR_WAVE", "1")
    engine._use_library_for_tests(os.path.join(os.path.dirname(emul_lib_path), lib))
    try:
        sc = scenes.make_scene(*meshgen.bumpy_torus(100, 60), 256)
        c, got, ref, qs = _event(engine, oracle, sc, 256 if lib == "libsurtr_emul.so" else 128)
    finally:
        engine._use_library_for_tests(None)
    assert c.status == 0 and c.n_frag == ref["frag_ids"].shape[0] > 80
    assert_event_equal(got, ref)
    assert np.array_equal(got["mesh_pos"], ref["mesh_pos"])       # same float program => bit-exact
    took, handed = int(qs[88]), int(qs[89])
    assert took + handed > 0
    if lib == "libsurtr_emul.so":
        assert took > 200, "the record clipper took %d pairs" % took       # (its capacities hold every band of this scene)
        # (one-kernel arrangement: handed on in place; split arrangement: the pairs k_prep_pairs knows to be irregular go straight
        #  to k_clip_pairs_catch, counted in slot 94)
        assert handed + int(qs[94]) > 0, "no pair with an in-plane vertex went to the general clipper"
    else:
        assert took + int(qs[94]) > 3 and handed > 10, (took, handed)      # (little room: many pairs run out of it and are redone by the general clipper)
        assert sum(int(qs[96 + r]) for r in (4, 7, 9, 10, 16)) > 5, "no pair ran out of room"


def test_hand_overs_nobody_polled_for_are_swept(emul_lib_path, oracle, monkeypatch):
    """The catcher beside the main kernel is an optimisation, not a dependency: with no workgroup polling (as when a profiler
    serialises the kernels and the catcher runs first), every pair the record clipper hands on is taken by the sweep launch."""
    from surtr_amd import engine
    monkeypatch.setenv("SURTR_WAVE", "1"); monkeypatch.setenv("SURTR_CATCH_POLL", "0")
    engine._use_library_for_tests(os.path.join(os.path.dirname(emul_lib_path), "libsurtr_emul_rec.so"))
    try:
        sc = scenes.make_scene(*meshgen.bumpy_torus(100, 60), 256)
        c, got, ref, qs = _event(engine, oracle, sc, 128)
    finally:
        engine._use_library_for_tests(None)
    assert c.status == 0 and c.n_frag == ref["frag_ids"].shape[0]
    assert_event_equal(got, ref)
    handed = int(qs[89])
    assert handed > 10 and int(qs[94]) >= handed, (handed, int(qs[94]))


def test_record_clipper_emulation_islands_and_empty_results(emul_engine, oracle, monkeypatch):
    """Two disjoint cubes in one piece (islands), and cells that keep nothing of the Mesh."""
    monkeypatch.setenv("SURTR_WAVE", "1")
    v, t = meshgen.bumpy_torus(60, 40)
    v2 = np.concatenate([v, v * np.float32(0.25) + np.float32([4, 0, 0])])
    t2 = np.concatenate([t, t + len(v)])
    sc = scenes.make_scene(v2, t2, 48)
    c, got, ref, qs = _event(emul_engine, oracle, sc, 48)
    assert c.status == 0
    assert_event_equal(got, ref)
    assert int(qs[88]) > 0


def test_record_clipper_off_is_the_general_clipper(emul_engine, oracle, monkeypatch):
    monkeypatch.setenv("SURTR_WAVE", "0")
    sc = scenes.make_scene(*meshgen.bumpy_torus(100, 60), 256)
    c, got, ref, qs = _event(emul_engine, oracle, sc, 64)
    assert_event_equal(got, ref)
    assert int(qs[88]) == 0 and int(qs[89]) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("wave", ["1", "0"])
def test_record_clipper_torus_4096_digest(gpu_engine, monkeypatch, wave):
    """BASELINE configs[3] with the record clipper forced on / off: the committed digests either way."""
    monkeypatch.setenv("SURTR_WAVE", wave)
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
    if wave == "1":
        assert int(qs[88]) > 2500, "the record clipper took only %d pairs" % int(qs[88])
        assert 0 < int(qs[89]) < 400
    else:
        assert int(qs[88]) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("cells,walk0", [(512, "0"), (512, "1"), (1024, "9")])
def test_record_clipper_small_blocks_and_walk_thresholds(gpu_engine, oracle, monkeypatch, cells, walk0):
    """Blocks of configs[3] that the engine would leave to the general clipper, forced through the record clipper, with the
    pointer jumping starting at the first / second / tenth walk step."""
    monkeypatch.setenv("SURTR_WAVE", "1")
    monkeypatch.setenv("SURTR_WWALK0", walk0)
    sc = scenes.torus_scene(4096)
    c, got, ref, qs = _event(gpu_engine, oracle, sc, cells, threads=16)
    assert c.status == 0
    assert_event_equal(got, ref)
    assert int(qs[88]) > 0


def test_record_clipper_big_bands_kernel_emulation(emul_lib_path, oracle, monkeypatch):
    """k_clip_pairs_wave_big (the record clipper with a whole CU's LDS, for the bands that leave the regular topology no room:
    cost classes 15..14) forced on, in the emulation build whose LDS topology is tiny -- every band is a "large" one there."""
    from surtr_amd import engine
    monkeypatch.setenv("SURTR_WAVE", "1")
    monkeypatch.setenv("SURTR_WAVE_BIG", "1")
    engine._use_library_for_tests(os.path.join(os.path.dirname(emul_lib_path), "libsurtr_emul_mid.so"))
    try:
        sc = scenes.make_scene(*meshgen.bumpy_torus(100, 60), 256)
        c, got, ref, qs = _event(engine, oracle, sc, 128)
    finally:
        engine._use_library_for_tests(None)
    assert c.status == 0
    assert_event_equal(got, ref)
    assert int(qs[88]) > 5 and int(qs[16 + 14]) + int(qs[16 + 15]) > 5, (int(qs[88]), qs[16:32].tolist())


@pytest.mark.gpu
@pytest.mark.parametrize("big", ["1", "0"])
def test_record_clipper_big_bands_kernel_gpu(gpu_engine, oracle, monkeypatch, big):
    """A 100 000-vertex torus x 4 096 cells (bands beyond the regular LDS topology are the rule): sampled cells against the
    oracle with the large bands through k_clip_pairs_wave_big / through k_clip_pairs_big."""
    monkeypatch.setenv("SURTR_WAVE_BIG", big)
    v, t = meshgen.bumpy_torus(500, 200)
    eng = gpu_engine.Engine(0)
    try:
        sc = scenes.make_scene(v, t, 4096, eng=eng)
        sc["mesh"] = eng.neighbors_from_mesh(v, t)[0]
        eng.upload_pieces([sc["mesh"]], [sc["convex"]]); eng.place_cells(sc["scale"], sc["translate"])
        c = eng.fracture_event(0, 4096)
        assert c.status == 0 and c.n_frag > 2500
        qs = eng.queue_stats()
        assert (int(qs[88]) > 1000) if big == "1" else True
        whole = eng.download()
        planes = oracle.place_cells(sc["v012"], sc["scale"], sc["translate"])
        ids = whole["frag_ids"]
        for cell in range(0, 4096, 256):
            ref = oracle.event([sc["mesh"]], [sc["convex"]], sc["face_off"], planes, refit=True, render=True, threads=1, cell_begin=cell, cell_end=cell + 1)
            ks = np.nonzero(ids[:, 0] == cell)[0]
            assert ks.shape[0] == ref["frag_ids"].shape[0]
            for j, k in enumerate(ks):
                a, b = fragment(whole, int(k), "mesh"), fragment(ref, j, "mesh")
                assert np.array_equal(a["off"], b["off"]) and np.array_equal(a["nbr"], b["nbr"]) and np.array_equal(a["pos"], b["pos"])
                # triangles too: the cap of a cell on this piece has up to ~90 vertices -- faces of 65..256 vertices are ear-clipped
                # by a wave with several vertices per lane (ear_clip_face_wave_k), the same triangles in the same order
                i0, i1 = int(whole["idx_off"][k]), int(whole["idx_off"][k + 1]); r0, r1 = int(ref["idx_off"][j]), int(ref["idx_off"][j + 1])
                assert np.array_equal(whole["idx"][i0:i1], ref["idx"][r0:r1])
    finally:
        eng.close()
