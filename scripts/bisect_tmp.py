import sys, os, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
from surtr_amd import engine as E, scenes
from oracle import oracle as O
from helpers import run_event
lib = sys.argv[1] if len(sys.argv) > 1 else None
if lib: E._use_library_for_tests(os.path.abspath(lib))
for name, sc in (("cube", scenes.cube_scene(8)), ("blob", scenes.blob_scene(64))):
    c, got, ref = run_event(E, O, sc, 2)
    same = np.array_equal(got["idx_off"], ref["idx_off"]) and np.array_equal(got["idx"], ref["idx"])
    print(lib, name, "idx equal:", same, "n_idx", c.n_idx, ref["idx"].shape[0], "n_failed", c.n_failed)
