"""Times one event per library variant (A/B of build-time knobs) -- exploratory."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from surtr_amd import engine as E, scenes as S
sc = S.torus_scene(4096)
if os.environ.get("SURTR_ACH"):
    _e = E.Engine(0); sc["convex"], _ = S.ach_convex(_e, sc["mesh"]["pos"]); _e.close()
for lib in sys.argv[1:]:
    E._use_library_for_tests(os.path.abspath(lib))
    eng = E.Engine(0)
    eng.upload_pieces([sc["mesh"]], [sc["convex"]]); eng.upload_pattern(sc["face_off"], sc["v012"]); eng.place_cells(sc["scale"], sc["translate"])
    c = eng.fracture_event(0, 4096)
    eng.set_profiling(True)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); c = eng.fracture_event(0, 4096); ts.append((time.perf_counter() - t0) * 1e3)
        kt = eng.kernel_times()
    print(lib, "frags", c.n_frag, "idx", c.n_idx, "status", c.status, "event ms %.2f" % min(ts), {k: round(v, 3) for k, v in kt.items()}, flush=True)
    eng.close()
