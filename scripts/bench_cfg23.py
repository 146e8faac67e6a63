"""Times BASELINE configs[1] and [2] (2 562-vertex blob x 64 / 1 024 cells) on the GPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from surtr_amd import engine as E, scenes as S
if os.environ.get('SURTR_LIB'):
    E._use_library_for_tests(os.path.abspath(os.environ['SURTR_LIB']))
eng = E.Engine(0)
for n in (64, 1024):
    sc = S.blob_scene(n)
    sc["convex"], _ = S.ach_convex(eng, sc["mesh"]["pos"])
    eng.upload_pieces([sc["mesh"]], [sc["convex"]]); eng.upload_pattern(sc["face_off"], sc["v012"]); eng.place_cells(sc["scale"], sc["translate"])
    c = eng.fracture_event(0, n)
    eng.set_profiling(True)
    ts = []
    for _ in range(8):
        t0 = time.perf_counter(); c = eng.fracture_event(0, n); ts.append((time.perf_counter() - t0) * 1e3)
    kt = eng.kernel_times()
    print("blob %d cells: %d fragments, %.3f ms per event (%.0f fragments/s); kernels %s" % (n, c.n_frag, min(ts), c.n_frag / min(ts) * 1e3, {k: round(v, 3) for k, v in kt.items() if v >= 0}), flush=True)
eng.close()
