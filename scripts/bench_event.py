"""Latency of ONE configs[3] event (min / median of 40) and its kernels' HIP-event times, for A/B runs of two builds:
SURTR_LIB=path/to/libsurtr_hip.so python scripts/bench_event.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from surtr_amd import engine as E, scenes as S
if os.environ.get("SURTR_LIB"): E._use_library_for_tests(os.path.abspath(os.environ["SURTR_LIB"]))
sc = S.torus_scene(4096)
eng = E.Engine(0)
sc["convex"], _ = S.ach_convex(eng, sc["mesh"]["pos"])
eng.upload_pieces([sc["mesh"]], [sc["convex"]]); eng.upload_pattern(sc["face_off"], sc["v012"]); eng.place_cells(sc["scale"], sc["translate"])
for _ in range(3): eng.fracture_event(0, 4096)
ts = []
for _ in range(40):
    t0 = time.perf_counter(); eng.place_cells(sc["scale"], sc["translate"]); eng.fracture_event(0, 4096); ts.append((time.perf_counter() - t0) * 1e3)
ts.sort()
eng.set_profiling(True)
acc = {}
for _ in range(10):
    eng.fracture_event(0, 4096)
    for k, v in eng.kernel_times().items():
        if v >= 0: acc.setdefault(k, []).append(v)
q = eng.queue_stats()
tag = " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("SURTR_")) or "default"
print("%s: event ms min %.3f median %.3f | record clipper took %d handed on %d %s | record images %d (given up on: %d) | kernels (median ms) %s" % (tag, ts[0], ts[len(ts) // 2],
      int(q[88]), int(q[89]), {i: int(q[96 + i]) for i in range(1, 20) if q[96 + i]}, int(q[91]), int(q[93]), {k: round(float(np.median(v)), 3) for k, v in acc.items() if np.median(v) > 0.02}), flush=True)
eng.close()
