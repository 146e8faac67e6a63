"""world_size-2 gloo test of the multi-GPU path: cells sharded in contiguous blocks, one all-gather of the
packed fragment blobs, deterministic concatenation.  Runs on CPU with the emulated kernels."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, emul_path, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from surtr_amd import engine, scenes
    from surtr_amd.multigpu import allgather_fragments
    engine._use_library_for_tests(emul_path)
    sc = scenes.blob_scene(64)
    eng = engine.Engine(0)
    eng.upload_pieces([sc["mesh"]], [sc["convex"]])
    eng.upload_pattern(sc["face_off"], sc["v012"])
    eng.place_cells(sc["scale"], sc["translate"])
    b, e = engine.cell_block(rank, world, sc["n_cells"])
    c = eng.fracture_event(b, e, flags=3)
    merged = allgather_fragments(eng, c, device="cpu")
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), **merged)
    eng.close()
    dist.destroy_process_group()


def test_two_ranks_equal_one(tmp_path, emul_lib_path, oracle):
    from surtr_amd import scenes
    from helpers import assert_event_equal
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, emul_lib_path, str(tmp_path)), nprocs=2, join=True)
    sc = scenes.blob_scene(64)
    planes = oracle.place_cells(sc["v012"], sc["scale"], sc["translate"])
    ref = oracle.event([sc["mesh"]], [sc["convex"]], sc["face_off"], planes, threads=4)
    for r in range(2):
        got = dict(np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)))
        assert_event_equal(got, ref)


def test_cell_blocks_cover_all_cells():
    from surtr_amd import engine
    for world in (1, 2, 3, 4, 8):
        for n in (1, 7, 64, 4096):
            blocks = [engine.cell_block(r, world, n) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))


def _refracture_inputs(engine, scenes):
    """First level on one rank's worth of work (every rank computes it: inputs are replicated), then the pair list."""
    sc = scenes.blob_scene(12)
    eng = engine.Engine(0)
    eng.upload_pieces([sc["mesh"]], [sc["convex"]])
    eng.upload_pattern(sc["face_off"], sc["v012"])
    eng.place_cells(sc["scale"], sc["translate"])
    eng.fracture_event(0, 12, flags=1)
    assert eng.pieces_from_event() > 6          # the first level's fragments are the second level's pieces, on the device
    first = eng.download()
    meshes, convexes = scenes.fragments_as_pieces(first)
    rs = scenes.refracture_scene(meshes, convexes, 4)
    eng.upload_pattern(rs["face_off"], rs["v012"])
    eng.place_cells_groups(rs["group_cell_off"], rs["scales"], rs["shifts"])
    return eng, rs


def _pair_worker(rank, world, port, emul_path, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from surtr_amd import engine, scenes
    from surtr_amd.multigpu import allgather_fragments
    engine._use_library_for_tests(emul_path)
    eng, rs = _refracture_inputs(engine, scenes)
    b, e = engine.pair_block(rank, world, rs["pair_cell"].shape[0])
    c = eng.fracture_pairs(rs["pair_cell"][b:e], rs["pair_piece"][b:e])
    merged = allgather_fragments(eng, c, device="cpu")
    np.savez(os.path.join(out_dir, "pairs_rank%d.npz" % rank), **merged)
    eng.close()
    dist.destroy_process_group()


def test_pair_sharded_refracture_equals_one_rank(tmp_path, emul_lib_path):
    """cfg5-style sharding: the (fragment, cell) pair list in contiguous fragment-major blocks over two ranks."""
    from surtr_amd import engine, scenes
    from helpers import assert_event_equal
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_pair_worker, args=(2, port, emul_lib_path, str(tmp_path)), nprocs=2, join=True)
    engine._use_library_for_tests(emul_lib_path)
    try:
        eng, rs = _refracture_inputs(engine, scenes)
        eng.fracture_pairs(rs["pair_cell"], rs["pair_piece"])
        ref = eng.download()
        eng.close()
    finally:
        engine._use_library_for_tests(None)
    assert ref["frag_ids"].shape[0] > 20
    for r in range(2):
        got = dict(np.load(os.path.join(str(tmp_path), "pairs_rank%d.npz" % r)))
        assert_event_equal(got, ref)


def test_pair_blocks_cover_the_list():
    from surtr_amd import engine
    for world in (1, 2, 3, 8):
        for n in (1, 5, 7488):
            blocks = [engine.pair_block(r, world, n) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n and all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))


def _balanced_worker(rank, world, port, emul_path, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from surtr_amd import engine, scenes
    from surtr_amd.multigpu import allgather_fragments
    engine._use_library_for_tests(emul_path)
    sc = scenes.blob_scene(64)
    eng = engine.Engine(0)
    eng.upload_pieces([sc["mesh"]], [sc["convex"]])
    eng.upload_pattern(sc["face_off"], sc["v012"])
    eng.place_cells(sc["scale"], sc["translate"])
    # every rank runs the whole event once and cuts the cells where the running cost passes rank / world of the total
    eng.fracture_event(0, 64)
    cuts = engine.balanced_blocks(eng.pair_costs(64), world)
    c = eng.fracture_event(cuts[rank], cuts[rank + 1])
    merged = allgather_fragments(eng, c, device="cpu")
    np.savez(os.path.join(out_dir, "bal_rank%d.npz" % rank), cuts=np.asarray(cuts), **merged)
    eng.close()
    dist.destroy_process_group()


def test_cost_balanced_blocks_merge_to_the_whole_event(tmp_path, emul_lib_path, oracle):
    """Strong sharding with cost-balanced CONTIGUOUS blocks: both ranks derive the same cuts from the
    same costs, and the blocks merged in rank order are the reference's cell-major event (Src/Surtr.cpp:2133-2146)."""
    from surtr_amd import scenes
    from helpers import assert_event_equal
    port = 33500 + (os.getpid() % 2000)
    mp.spawn(_balanced_worker, args=(2, port, emul_lib_path, str(tmp_path)), nprocs=2, join=True)
    sc = scenes.blob_scene(64)
    planes = oracle.place_cells(sc["v012"], sc["scale"], sc["translate"])
    ref = oracle.event([sc["mesh"]], [sc["convex"]], sc["face_off"], planes, threads=4)
    cuts = None
    for r in range(2):
        d = dict(np.load(os.path.join(str(tmp_path), "bal_rank%d.npz" % r)))
        c = d.pop("cuts")
        assert cuts is None or np.array_equal(cuts, c)
        cuts = c
        assert_event_equal(d, ref)
    assert cuts[0] == 0 and cuts[-1] == 64 and 0 < cuts[1] < 64


def test_balanced_blocks_are_contiguous_and_cover():
    from surtr_amd import engine
    rng = np.random.default_rng(7)
    for world in (1, 2, 3, 8):
        for n in (1, 5, 64, 4096):
            costs = rng.integers(1, 1000, n)
            cuts = engine.balanced_blocks(costs, world)
            assert len(cuts) == world + 1 and cuts[0] == 0 and cuts[-1] == n and all(cuts[i] <= cuts[i + 1] for i in range(world))
    # heavy cells in front: the first block is short
    cuts = engine.balanced_blocks([100] * 8 + [1] * 56, 2)
    assert cuts[1] < 16
