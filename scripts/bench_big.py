"""A mesh ten times configs[3] (bumpy torus 1000 x 500 = 500 000 vertices / 1 000 000 triangles) x 4 096 cells: does the
engine size its scratch sensibly, how long does the event take, is it still the oracle's event (sampled cells).
Usage: python scripts/bench_big.py [nu] [nv] [cells] [sample]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from surtr_amd import engine as E, scenes as S, meshgen as G
nu = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
nv = int(sys.argv[2]) if len(sys.argv) > 2 else 500
C = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
sample = int(sys.argv[4]) if len(sys.argv) > 4 else 64
eng = E.Engine(0)
t0 = time.perf_counter()
v, t = G.bumpy_torus(nu, nv)
sc = S.make_scene(v, t, C, eng=eng)
sc["mesh"] = eng.neighbors_from_mesh(v, t)[0]
sc["convex"], _ = S.ach_convex(eng, sc["mesh"]["pos"])
print("scene: %d vertices, %d triangles, %d cells (%.1f s)" % (len(v), len(t), C, time.perf_counter() - t0), flush=True)
eng.upload_pieces([sc["mesh"]], [sc["convex"]]); eng.place_cells(sc["scale"], sc["translate"])
free0, total = torch.cuda.mem_get_info()
c = eng.fracture_event(0, C)
free1, _ = torch.cuda.mem_get_info()
print("first event: status %d, %d fragments, %d mesh vertices, %d indices; device memory in use %.1f GB of %.0f GB (event buffers %.1f GB)"
      % (c.status, c.n_frag, c.mesh_verts, c.n_idx, (total - free1) / 2**30, total / 2**30, (free0 - free1) / 2**30), flush=True)
eng.set_profiling(True)
ts = []
for _ in range(5):
    t0 = time.perf_counter(); c = eng.fracture_event(0, C); ts.append((time.perf_counter() - t0) * 1e3)
print("event %.2f ms (%.0f fragments/s); kernels" % (min(ts), c.n_frag / min(ts) * 1e3), {k: round(x, 3) for k, x in eng.kernel_times().items() if x > 0}, flush=True)
if sample:
    from oracle import oracle as O
    from helpers import assert_event_equal
    cells = np.arange(0, C, max(1, C // sample), dtype=np.int64)[:sample]
    planes = O.place_cells(sc["v012"], sc["scale"], sc["translate"])
    t0 = time.perf_counter()
    ok = 0
    for cell in cells:
        cc = eng.fracture_event(int(cell), int(cell) + 1)
        got = eng.download()
        ref = O.event([sc["mesh"]], [sc["convex"]], sc["face_off"], planes, refit=True, render=True, threads=1, cell_begin=int(cell), cell_end=int(cell) + 1)
        assert_event_equal(got, ref); ok += 1
    print("oracle check: %d of %d sampled cells equal (%.1f s)" % (ok, len(cells), time.perf_counter() - t0), flush=True)
eng.close()
