"""Edge cases of the round-2 entry points (empty / tiny / degenerate inputs, call-order errors).  `check_*` run on the emulation
here and on the MI355X in tests/test_gpu_parity.py."""
import numpy as np
import pytest

from helpers import assert_event_equal
from surtr_amd import scenes


def check_tiny_cell_counts(E):
    """One seed = the whole unit box; two and three seeds; all equal to the host builder."""
    import test_build_cells as bc
    for n in (1, 2, 3):
        bc.check_cells(E, n)
    eng = E.Engine(0)
    try:
        nf, nfv = eng.build_cells(np.zeros((1, 3)))
        assert (nf, nfv) == (6, 24)
        with pytest.raises(E.SurtrError) as e:
            eng.build_cells(np.zeros((0, 3)), np.array([0, 0], np.uint32))
        assert e.value.code == E.E_INVALID
    finally:
        eng.close()


def check_call_order_and_bad_arguments(E, oracle):
    box = oracle.unit_box()
    eng = E.Engine(0)
    try:
        for call in (lambda: eng.event_regroup(), lambda: eng.pieces_from_event(), lambda: eng.event_triangulate(), lambda: eng.event_refit(),
                     lambda: eng.transform_pieces(np.eye(4, dtype=np.float32)[None])):
            with pytest.raises(E.SurtrError) as e:
                call()
            assert e.value.code == E.E_STATE
        eng.upload_pieces([box], [box])
        with pytest.raises(E.SurtrError) as e:
            eng.transform_pieces(np.zeros((2, 4, 4), np.float32))          # one matrix per piece
        assert e.value.code == E.E_INVALID
        tri = {"pos": box["pos"][:3], "off": np.array([0, 2, 4, 6], np.uint32), "nbr": np.array([1, 2, 2, 0, 0, 1], np.int32)}
        with pytest.raises(E.SurtrError) as e:
            eng.load_fragments([tri], [box])                               # fewer than four vertices is no solid
        assert e.value.code == E.E_INVALID
        eng.load_fragments([box], [box])
        with pytest.raises(E.SurtrError) as e:
            eng.pieces_from_event(np.zeros(1, np.uint8))                   # nothing kept
        assert e.value.code == E.E_INVALID
        assert eng.pieces_from_event() == 1
        got = eng.download()                                               # the loaded fragment is still there
        assert got["frag_ids"].shape[0] == 1 and np.array_equal(got["mesh_pos"], box["pos"])
    finally:
        eng.close()


def check_projective_transform(E, oracle):
    """Poly::Transform divides by w (XMVector3TransformCoord): a matrix with a projective row."""
    import test_solid_ops as ops
    box = oracle.unit_box()
    w = np.eye(4, dtype=np.float32); w[3] = [0.1, -0.05, 0.2, 1.5]; w[0, 3] = 2.0
    eng = E.Engine(0)
    try:
        eng.upload_pieces([box], [box])
        eng.transform_pieces(w[None])
        eng.upload_planes(np.array([0, 1], np.uint32), np.array([[0, 0, 1, -100]], np.float32))     # a plane that cuts nothing
        eng.fracture_event(0, 1, flags=0)
        got = eng.download()
    finally:
        eng.close()
    assert np.array_equal(got["mesh_pos"], oracle.transform(box, w)["pos"])


def check_whole_piece_survives_and_vanishes(E, oracle):
    """A cell that contains the whole piece returns it unchanged; a cell that misses it returns nothing; both in one event."""
    box = oracle.unit_box()
    planes = np.array([[1, 0, 0, -5], [-1, 0, 0, -5],          # cell 0: |x| <= 5
                       [1, 0, 0, 7], [-1, 0, 0, -9]], np.float32)      # cell 1: -9 <= x <= -7
    fo = np.array([0, 2, 4], np.uint32)
    eng = E.Engine(0)
    try:
        eng.upload_pieces([box], [box]); eng.upload_planes(fo, planes)
        c = eng.fracture_event(0, 2, flags=3)
        got = eng.download()
    finally:
        eng.close()
    ref = oracle.event([box], [box], fo, planes)
    assert c.n_frag == 1 and np.array_equal(got["frag_ids"], [[0, 0, 0]])
    assert_event_equal(got, ref)
    assert np.array_equal(got["mesh_pos"], box["pos"]) and got["idx"].shape[0] == 36


CASES = [check_tiny_cell_counts, check_call_order_and_bad_arguments, check_projective_transform, check_whole_piece_survives_and_vanishes]


@pytest.mark.parametrize("case", CASES, ids=lambda f: f.__name__)
def test_edge_cases_emulated(emul_engine, oracle, case):
    import inspect
    case(emul_engine, oracle) if len(inspect.signature(case).parameters) == 2 else case(emul_engine)
