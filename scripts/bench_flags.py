"""Times the cfg4 event with refit only / triangulation only / both / neither (per-kernel HIP-event times) -- exploratory."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from surtr_amd import engine as E, scenes as S
sc = S.torus_scene(4096)
_e = E.Engine(0); sc["convex"], _ = S.ach_convex(_e, sc["mesh"]["pos"]); _e.close()
if len(sys.argv) > 1:
    E._use_library_for_tests(os.path.abspath(sys.argv[1]))
eng = E.Engine(0)
eng.upload_pieces([sc["mesh"]], [sc["convex"]]); eng.upload_pattern(sc["face_off"], sc["v012"]); eng.place_cells(sc["scale"], sc["translate"])
for flags in ((E.EVT_REFIT | E.EVT_RENDER,) if os.environ.get("SURTR_REFIT_WG_BOTH") else (E.EVT_RENDER, E.EVT_REFIT | E.EVT_RENDER) if os.environ.get("SURTR_FACES_WG") else (0, E.EVT_REFIT, E.EVT_RENDER, E.EVT_REFIT | E.EVT_RENDER)):
    c = eng.fracture_event(0, 4096, flags=flags)
    eng.set_profiling(True)
    ts = []
    for _ in range(6):
        t0 = time.perf_counter(); c = eng.fracture_event(0, 4096, flags=flags); ts.append((time.perf_counter() - t0) * 1e3)
        kt = eng.kernel_times()
    eng.set_profiling(False)
    ts2 = []
    for _ in range(6):
        t0 = time.perf_counter(); c = eng.fracture_event(0, 4096, flags=flags); ts2.append((time.perf_counter() - t0) * 1e3)
    print("flags", flags, "event ms %.3f (profiling on) %.3f (off)" % (min(ts), min(ts2)), {k: round(v, 3) for k, v in kt.items() if v > 0}, flush=True)
eng.close()
