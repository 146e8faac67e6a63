"""Times BASELINE configs[4] (recursive refracture: 256 first-level fragments x 32 cells each) on the GPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from surtr_amd import engine as E, scenes as S, meshgen as G
if os.environ.get('SURTR_LIB'):
    E._use_library_for_tests(os.path.abspath(os.environ['SURTR_LIB']))
sc = S.make_scene(*G.bumpy_torus(), 256)
eng = E.Engine(0)
eng.upload_pieces([sc["mesh"]], [sc["convex"]]); eng.upload_pattern(sc["face_off"], sc["v012"]); eng.place_cells(sc["scale"], sc["translate"])
t0 = time.perf_counter(); c = eng.fracture_event(0, 256, flags=1); t1 = time.perf_counter()
print("level 1: 256 cells, %d fragments, %.2f ms (first call, allocations included)" % (c.n_frag, (t1 - t0) * 1e3))
first = eng.download()
meshes, convexes = S.fragments_as_pieces(first)
keep = [i for i, m in enumerate(meshes) if m["pos"].shape[0] >= 4 and np.diff(m["off"].astype(np.int64)).min() >= 3 and convexes[i]["pos"].shape[0] >= 4]
meshes, convexes = [meshes[i] for i in keep], [convexes[i] for i in keep]
rs = S.refracture_scene(meshes, convexes, 32)
eng.upload_pieces(meshes, convexes); eng.upload_pattern(rs["face_off"], rs["v012"]); eng.place_cells_groups(rs["group_cell_off"], rs["scales"], rs["shifts"])
eng.set_profiling(True)
ts = []
for _ in range(6):
    t0 = time.perf_counter(); c = eng.fracture_pairs(rs["pair_cell"], rs["pair_piece"], flags=3); ts.append((time.perf_counter() - t0) * 1e3)
kt = eng.kernel_times()
print("level 2: %d pieces x 32 cells = %d pairs, %d fragments, %d mesh verts: %.2f ms per event; kernels %s" % (
    len(meshes), rs["pair_cell"].shape[0], c.n_frag, c.mesh_verts, min(ts), {k: round(v, 3) for k, v in kt.items() if v >= 0}))
eng.close()
