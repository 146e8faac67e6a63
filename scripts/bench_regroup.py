"""The DoFracture chain (Src/Surtr.cpp:1885-1959) on the device for the level-2 event of BASELINE configs[4]: 234 pieces x 32 cells
each -- event without refit -> surtr_event_regroup (bind sets, MergeOutOfImpact off, HandleConvexIsland) -> surtr_event_refit.
Wall time of each phase (host clock around the C-ABI call, device idle before it), rounds of the label propagation."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from surtr_amd import engine as E, scenes as S, meshgen as G
N1, N2 = 256, 32
eng = E.Engine(0)
sc = S.make_scene(*G.bumpy_torus(), N1, eng=eng)
eng.upload_pieces([sc["mesh"]], [sc["convex"]])
eng.upload_pattern(sc["face_off"], sc["v012"]); eng.place_cells(sc["scale"], sc["translate"])
c1 = eng.fracture_event(0, N1, flags=1)
n = eng.pieces_from_event()
go = np.arange(0, n * N2 + 1, N2, dtype=np.uint32)
seeds = np.concatenate([S.uniform_seeds(N2, S.SEED + p) for p in range(n)])
eng.build_cells(seeds, go); eng.place_cells_in_pieces(go)
pc, pp = np.arange(n * N2, dtype=np.uint32), np.repeat(np.arange(n, dtype=np.uint32), N2)
rows = []
for rep in range(6):
    t0 = time.perf_counter(); c2 = eng.fracture_pairs(pc, pp, flags=0); t1 = time.perf_counter()
    co, cp = eng.event_regroup(); t2 = time.perf_counter()
    eng.event_refit(); c3 = eng.event_counts(); t3 = time.perf_counter()
    rows.append(((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
r = np.array(rows[1:])
print("%d pieces x %d cells: %d pairs -> %d fragments -> %d compounds | event (no refit) %.2f ms, regroup %.2f ms, refit %.2f ms (median of %d)" % (
    n, N2, pc.shape[0], c2.n_frag, co.shape[0] - 1, *np.median(r, 0), r.shape[0]))
eng.close()
