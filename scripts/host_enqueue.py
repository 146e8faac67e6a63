"""How long the HOST takes to enqueue one configs[3] event (place + event + pack, no synchronisation): if this is close to the
per-step time with several events in flight, the steps are bound by the enqueueing thread, not by the GPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from surtr_amd import engine as E, scenes as S
sc = S.torus_scene(4096)
engs = []
for k in range(3):
    e = E.Engine(0); st = torch.cuda.Stream(); e.set_stream(st.cuda_stream); engs.append((e, st))
    if k == 0: sc["convex"], _ = S.ach_convex(e, sc["mesh"]["pos"])
    e.upload_pieces([sc["mesh"]], [sc["convex"]]); e.upload_pattern(sc["face_off"], sc["v012"]); e.place_cells(sc["scale"], sc["translate"])
    c = e.fracture_event(0, 4096)
cap = E.blob_bytes(c) + 4096
blobs = [torch.zeros(cap, dtype=torch.uint8, device="cuda") for _ in engs]
def step(i):
    e, _ = engs[i % 3]
    e.place_cells(sc["scale"], sc["translate"]); e.fracture_event_async(0, 4096); e.pack_dev(blobs[i % 3].data_ptr(), cap)
for i in range(6): step(i)
torch.cuda.synchronize()
ts = []
t0 = time.perf_counter()
for i in range(60):
    a = time.perf_counter(); step(i); ts.append((time.perf_counter() - a) * 1e3)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("host enqueue per event: median %.3f ms, mean %.3f ms (60 events enqueued in %.2f ms, drained after %.2f ms more)" % (np.median(ts), np.mean(ts), (t1 - t0) * 1e3, (t2 - t1) * 1e3))
