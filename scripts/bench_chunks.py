"""ONE configs[3] event cut into E contiguous cell blocks that E engines (contexts) on E streams of the same GPU run side by side:
what a chunked pipeline inside one event would buy (the kernels of one block fill the idle tails of the other's).  Prints the
latency of the whole event per E.  Usage: python scripts/bench_chunks.py [E ...]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from surtr_amd import engine as E, scenes as S
Es = [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4]
sc = S.torus_scene(4096)
boot = E.Engine(0)
sc["convex"], _ = S.ach_convex(boot, sc["mesh"]["pos"])
boot.upload_pieces([sc["mesh"]], [sc["convex"]]); boot.upload_pattern(sc["face_off"], sc["v012"]); boot.place_cells(sc["scale"], sc["translate"])
boot.fracture_event(0, 4096)
costs = boot.pair_costs(4096)
boot.close()
for n in Es:
    cuts = E.balanced_blocks(costs, n) if n > 1 else [0, 4096]
    engs, streams = [], []
    for k in range(n):
        st = torch.cuda.Stream()
        e = E.Engine(0); e.set_stream(st.cuda_stream)
        e.upload_pieces([sc["mesh"]], [sc["convex"]]); e.upload_pattern(sc["face_off"], sc["v012"]); e.place_cells(sc["scale"], sc["translate"])
        e.fracture_event(cuts[k], cuts[k + 1])
        engs.append(e); streams.append(st)
    torch.cuda.synchronize()
    ts = []
    for r in range(12):
        t0 = time.perf_counter()
        for k, e in enumerate(engs):
            e.place_cells(sc["scale"], sc["translate"])
            e.fracture_event_async(cuts[k], cuts[k + 1])
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    print("E=%d blocks %s: event ms min %.3f median %.3f" % (n, cuts, ts[0], ts[len(ts) // 2]), flush=True)
    for e in engs:
        e.close()
