"""Times the cell blocks a rank of an N-way strong-sharded configs[3] event would run (one GPU, block by block): equal-sized
blocks, or -- with --balanced -- contiguous blocks cut where the running cost estimate of one whole event
(surtr_event_pair_costs) passes r / N of the total.  Prints the slowest block and the predicted N-rank speed-up over the whole
event on one GPU (a prediction from one GPU: the all-gather and the other ranks' jitter are not in it).
Usage: python scripts/bench_slices.py N [--balanced]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from surtr_amd import engine as E, scenes as S
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
balanced = "--balanced" in sys.argv
sc = S.torus_scene(4096)
if os.environ.get("SURTR_LIB"): E._use_library_for_tests(os.path.abspath(os.environ["SURTR_LIB"]))
eng = E.Engine(0)
sc["convex"], _ = S.ach_convex(eng, sc["mesh"]["pos"])
eng.upload_pieces([sc["mesh"]], [sc["convex"]]); eng.upload_pattern(sc["face_off"], sc["v012"]); eng.place_cells(sc["scale"], sc["translate"])
eng.fracture_event(0, 4096)
whole = []
for _ in range(5):
    t0 = time.perf_counter(); eng.fracture_event(0, 4096); whole.append((time.perf_counter() - t0) * 1e3)
cuts = E.balanced_blocks(eng.pair_costs(4096), N) if balanced else [E.cell_block(r, N, 4096)[0] for r in range(N)] + [4096]
eng.set_profiling(True)
slowest = 0.0
for r in range(N):
    cb, ce = cuts[r], cuts[r + 1]
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); c = eng.fracture_event(cb, ce); ts.append((time.perf_counter() - t0) * 1e3)
    kt = eng.kernel_times()
    slowest = max(slowest, min(ts))
    print("rank %d of %d: cells [%d,%d) frags %d event ms %.2f" % (r, N, cb, ce, c.n_frag, min(ts)), {k: round(v, 3) for k, v in kt.items() if v >= 0}, flush=True)
print("%s blocks x%d: whole event %.2f ms, slowest block %.2f ms -> predicted speed-up %.2fx" % ("cost-balanced" if balanced else "equal-sized", N, min(whole), slowest, min(whole) / slowest))
eng.close()
