"""Replays one case of scripts/fuzz_gpu.py (same RNG stream) on a chosen library (default: the CPU emulation).
Usage: python scripts/fuzz_repro.py <seed> <case> [lib.so]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import numpy as np
from oracle import oracle as O
from surtr_amd import engine as E, scenes as S
from helpers import assert_event_equal
import fuzz_gpu as F

seed, case = int(sys.argv[1]), int(sys.argv[2])
lib = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "tests", "emul", "libsurtr_emul.so")
E._use_library_for_tests(lib)
rng = np.random.default_rng(seed)
eng = E.Engine(0)
for k in range(case + 1):
    sc, kind = F.random_scene(rng)
    use_ach = bool(rng.integers(0, 2))
    if use_ach and k == case:
        sc["convex"], _ = S.ach_convex(eng, sc["mesh"]["pos"])
    flags = int(rng.choice([0, 1, 2, 3, 3, 3]))
    scale = sc["scale"] * np.float32(rng.uniform(0.6, 2.2)); shift = sc["translate"] + (rng.uniform(-0.3, 0.3, 3) * sc["scale"]).astype(np.float32)
print("case", case, "kind", kind, "V", sc["mesh"]["pos"].shape[0], "cells", sc["n_cells"], "ach", use_ach, "flags", flags)
eng.upload_pieces([sc["mesh"]], [sc["convex"]]); eng.upload_pattern(sc["face_off"], sc["v012"]); eng.place_cells(scale, shift)
planes = O.place_cells(sc["v012"], scale, shift)
ref = O.event([sc["mesh"]], [sc["convex"]], sc["face_off"], planes, refit=bool(flags & 1), render=bool(flags & 2), threads=8)
print("oracle fragments", ref["frag_ids"].shape[0])
try:
    c = eng.fracture_event(0, sc["n_cells"], flags=flags)
    print("status", c.status, "frags", c.n_frag)
    assert_event_equal(eng.download(), ref, render=bool(flags & 2))
    print("MATCH")
except Exception as e:
    print("FAIL", repr(e)[:300])
    # which pair?  one cell at a time
    for cell in range(sc["n_cells"]):
        try:
            eng.fracture_event(cell, cell + 1, flags=flags)
        except Exception as e2:
            print("  cell", cell, "fails:", repr(e2)[:120])
            np.savez_compressed(os.path.join(ROOT, "gpurun_out", "repro_case.npz"), mesh_pos=sc["mesh"]["pos"], mesh_off=sc["mesh"]["off"], mesh_nbr=sc["mesh"]["nbr"],
                                conv_pos=sc["convex"]["pos"], conv_off=sc["convex"]["off"], conv_nbr=sc["convex"]["nbr"],
                                planes=planes[sc["face_off"][cell]:sc["face_off"][cell + 1]], cell=cell)
            break
