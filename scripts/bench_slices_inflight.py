"""The cell blocks of an N-way strong-sharded configs[3] event with E events in flight per GPU, as bench.py --gpus N runs them
(E contexts on E streams take the steps in turn): ms per step of every block on ONE GPU, block by block, and the predicted
N-rank throughput relative to the whole event with the same E (a prediction: all-gather and the other ranks' jitter are not in it).
Usage: python scripts/bench_slices_inflight.py N [E]"""
import sys, os, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")      # as bench.py: a hardware queue per stream
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from surtr_amd import engine as E_, scenes as S
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
E = int(sys.argv[2]) if len(sys.argv) > 2 else 6      # bench.py's IN_FLIGHT_DEFAULT
sc = S.torus_scene(4096)
engs, streams = [], []
for k in range(E):
    st = torch.cuda.Stream()
    e = E_.Engine(0); e.set_stream(st.cuda_stream); e.set_events_in_flight(E)
    if k == 0: sc["convex"], _ = S.ach_convex(e, sc["mesh"]["pos"])
    e.upload_pieces([sc["mesh"]], [sc["convex"]]); e.upload_pattern(sc["face_off"], sc["v012"]); e.place_cells(sc["scale"], sc["translate"])
    engs.append(e); streams.append(st)
engs[0].fracture_event(0, 4096)
cuts = E_.balanced_blocks(engs[0].pair_costs(4096), N)

def ms_per_step(cb, ce, steps=45):
    for e in engs: e.fracture_event(cb, ce)
    torch.cuda.synchronize()
    for i in range(2 * E): engs[i % E].place_cells(sc["scale"], sc["translate"]); engs[i % E].fracture_event_async(cb, ce)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps): engs[i % E].place_cells(sc["scale"], sc["translate"]); engs[i % E].fracture_event_async(cb, ce)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / steps

whole = ms_per_step(0, 4096)
worst = 0.0
for r in range(N):
    t = ms_per_step(cuts[r], cuts[r + 1])
    worst = max(worst, t)
    print("rank %d of %d: cells [%d,%d) %.3f ms per step with %d in flight" % (r, N, cuts[r], cuts[r + 1], t, E), flush=True)
print("x%d, %d in flight: whole event %.3f ms per step, slowest block %.3f ms per step -> predicted speed-up %.2fx" % (N, E, whole, worst, whole / worst))
for e in engs: e.close()
