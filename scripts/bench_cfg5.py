"""BASELINE configs[4] (recursive refracture: 256 first-level cells, every fragment re-split into 32 cells) end to end on the
GPU, the first level's fragments never leaving HBM:
    level 1 event -> surtr_pieces_from_event -> surtr_build_cells (one Voronoi diagram per fragment) -> placement per
    fragment -> level 2 event over the (fragment, cell) pair list.
Prints the wall time of the whole chain next to the two events alone."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from surtr_amd import engine as E, scenes as S, meshgen as G
if os.environ.get('SURTR_LIB'):
    E._use_library_for_tests(os.path.abspath(os.environ['SURTR_LIB']))
N1, N2 = 256, 32
eng = E.Engine(0)
sc = S.make_scene(*G.bumpy_torus(), N1, eng=eng)
pat1 = (sc["face_off"], sc["v012"])


def level1():
    eng.upload_pattern(*pat1); eng.place_cells(sc["scale"], sc["translate"])
    return eng.fracture_event(0, N1, flags=1)


SEEDS = np.concatenate([S.uniform_seeds(N2, S.SEED + p) for p in range(N1 + 64)])      # inputs of level 2, drawn once


def between(c1):
    """first level's fragments -> pieces; one Voronoi diagram per piece, placed over its box: all on the device"""
    n = eng.pieces_from_event()              # keeps every fragment that is a solid
    go = np.arange(0, n * N2 + 1, N2, dtype=np.uint32)
    eng.build_cells(SEEDS[:n * N2], go)
    eng.place_cells_in_pieces(go)
    return n, np.arange(n * N2, dtype=np.uint32), np.repeat(np.arange(n, dtype=np.uint32), N2)


eng.upload_pieces([sc["mesh"]], [sc["convex"]])
for rep in range(4):
    eng.upload_pieces([sc["mesh"]], [sc["convex"]])
    t0 = time.perf_counter(); c1 = level1(); t1 = time.perf_counter()
    n, pc, pp = between(c1); t2 = time.perf_counter()
    c2 = eng.fracture_pairs(pc, pp, flags=3); t3 = time.perf_counter()
    print("rep %d: level 1 %d cells -> %d fragments %.2f ms | fragments -> pieces + %d x %d cells built and placed %.2f ms | level 2 %d pairs -> %d fragments %.2f ms | chain %.2f ms" % (
        rep, N1, c1.n_frag, (t1 - t0) * 1e3, n, N2, (t2 - t1) * 1e3, pc.shape[0], c2.n_frag, (t3 - t2) * 1e3, (t3 - t0) * 1e3), flush=True)
eng.set_profiling(True)
ts = []
for _ in range(6):
    t0 = time.perf_counter(); c2 = eng.fracture_pairs(pc, pp, flags=3); ts.append((time.perf_counter() - t0) * 1e3)
kt = eng.kernel_times()
print("level 2 alone: %d pairs, %d fragments, %d mesh verts: %.2f ms per event; kernels %s" % (
    pc.shape[0], c2.n_frag, c2.mesh_verts, min(ts), {k: round(v, 3) for k, v in kt.items() if v >= 0}))
print("upload stats of the last pieces_from_event: %.3f ms host, %d allocations" % eng.upload_stats())
eng.close()
