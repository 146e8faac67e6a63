"""surtr_build_cells (Voronoi cells on the device, row A2) against the ORACLE's cells (oracle.voronoi_cells: the canonical
cell of DESIGN section 5 built as a polygon soup -- structure equal, coordinates to 1e-12; Src/Surtr.cpp:2003-2070 with voro++
replaced by the canonical order) and against the host builder surtr_voronoi_cells (bit for bit: same float program).
`check_cells` runs on the emulation here and on the MI355X in test_gpu_parity.py."""
import numpy as np
import pytest

from surtr_amd import scenes


def check_cells(E, n, groups=1, oracle=None):
    eng = E.Engine(0)
    try:
        if groups == 1:
            seeds = scenes.uniform_seeds(n)
            go = None
            parts = [seeds]
            refs = [E.voronoi_cells(seeds)]
        else:
            parts = [scenes.uniform_seeds(n, scenes.SEED + g) for g in range(groups)]
            seeds = np.concatenate(parts)
            go = np.arange(0, n * groups + 1, n, dtype=np.uint32)
            refs = [E.voronoi_cells(p) for p in parts]
        nf, nfv = eng.build_cells(seeds, go)
        got = eng.download_cells()
    finally:
        eng.close()
    f0 = v0 = c0 = 0
    for gi, ref in enumerate(refs):
        k, kv, kc = ref["face_gen"].shape[0], ref["verts"].shape[0], ref["cell_face_off"].shape[0] - 1
        assert np.array_equal(got["cell_face_off"][c0:c0 + kc + 1] - f0, ref["cell_face_off"])
        assert np.array_equal(got["face_gen"][f0:f0 + k], ref["face_gen"])
        assert np.array_equal(got["face_vert_off"][f0:f0 + k + 1] - v0, ref["face_vert_off"])
        assert np.array_equal(got["verts"][v0:v0 + kv], ref["verts"].reshape(-1, 3))          # doubles, bit for bit
        _, v012 = E.pattern_from_cells(ref)
        assert np.array_equal(got["v012"][f0:f0 + k], v012)
        if oracle is not None and (groups == 1 or gi % 29 == 0):
            # the checker proper: the oracle's own construction of the same canonical cells
            o = oracle.voronoi_cells(parts[gi])
            assert np.array_equal(got["cell_face_off"][c0:c0 + kc + 1] - f0, o["cell_face_off"])
            assert np.array_equal(got["face_gen"][f0:f0 + k], o["face_gen"])
            assert np.array_equal(got["face_vert_off"][f0:f0 + k + 1] - v0, o["face_vert_off"])
            assert np.abs(got["verts"][v0:v0 + kv] - o["verts"].reshape(-1, 3)).max() < 1e-12
        f0 += k; v0 += kv; c0 += kc
    assert (nf, nfv) == (f0, v0)


@pytest.mark.parametrize("n,groups", [(8, 1), (64, 1), (300, 1), (32, 5)])
def test_build_cells_emulated(emul_engine, oracle, n, groups):
    check_cells(emul_engine, n, groups, oracle)


def test_built_pattern_drives_an_event(emul_engine, oracle):
    """The cells built on the device are the context's pattern: an event on them equals the oracle's on the host cells."""
    sc = scenes.cube_scene(8)
    eng = emul_engine.Engine(0)
    try:
        eng.upload_pieces([sc["mesh"]], [sc["convex"]])
        eng.build_cells(sc["seeds"])
        eng.place_cells(sc["scale"], sc["translate"])
        eng.fracture_event(0, 8)
        got = eng.download()
    finally:
        eng.close()
    from helpers import assert_event_equal
    planes = oracle.place_cells(sc["v012"], sc["scale"], sc["translate"])
    assert_event_equal(got, oracle.event([sc["mesh"]], [sc["convex"]], sc["face_off"], planes))
