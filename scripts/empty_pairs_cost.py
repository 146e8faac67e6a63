"""How much of a configs[3] event is spent on pairs that yield no fragment (cells outside the solid but inside its Convex)?
Times the event restricted to the cells that produce fragments, and to those that do not."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from surtr_amd import engine as E, scenes as S
sc = S.torus_scene(4096)
eng = E.Engine(0)
sc["convex"], _ = S.ach_convex(eng, sc["mesh"]["pos"])
eng.upload_pieces([sc["mesh"]], [sc["convex"]]); eng.upload_pattern(sc["face_off"], sc["v012"]); eng.place_cells(sc["scale"], sc["translate"])
c = eng.fracture_event(0, 4096)
ids = eng.download()["frag_ids"]
full = np.unique(ids[:, 0]).astype(np.uint32)
empty = np.setdiff1d(np.arange(4096, dtype=np.uint32), full)
eng.set_profiling(True)
for name, cells in (("all cells", np.arange(4096, dtype=np.uint32)), ("cells with fragments", full), ("cells without", empty)):
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); c = eng.fracture_pairs(cells, np.zeros_like(cells)); ts.append((time.perf_counter() - t0) * 1e3)
    kt = eng.kernel_times()
    print("%-22s %5d pairs -> %5d fragments: %.2f ms  %s" % (name, cells.shape[0], c.n_frag, min(ts), {k: round(v, 3) for k, v in kt.items() if v > 0.02}), flush=True)
eng.close()
