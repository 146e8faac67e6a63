#!/usr/bin/env python3
"""How many pairs of a BASELINE configs[3] event the record clipper (wave_clip.h) takes and how many it hands on (by rule), the pairs
per cost class and the kernel times.  Usage: python scripts/wave_stats.py [cells] [nu nv]   (nu x nv: another torus, e.g. 500 200)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from surtr_amd import engine, scenes, meshgen

cells = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
eng = engine.Engine(0)
mesh = meshgen.bumpy_torus(int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else meshgen.bumpy_torus()
sc = scenes.mesh_scene(*mesh, eng=eng)
eng.build_cells(scenes.uniform_seeds(cells, scenes.SEED))
sc["convex"], _ = scenes.ach_convex(eng, sc["mesh"]["pos"])
eng.upload_pieces([sc["mesh"]], [sc["convex"]])
eng.place_cells(sc["scale"], sc["translate"])
eng.set_profiling(True)
for it in range(3):
    c = eng.fracture_event(0, cells)
    q = eng.queue_stats()
    t = eng.kernel_times()
print("fragments", c.n_frag, "wave took", q[88], "handed on", q[89], "retry list", q[64])
print("by rule:", {i: int(q[96 + i]) for i in range(1, 20) if q[96 + i]})
print("pairs per cost class (clip queue):", {c: int(q[16 + c]) for c in range(16) if q[16 + c]})
for j in range(1):
    o = 96 + 21
    if o + 8 <= len(q): print("site 10/16 sample (n, k, rtop, M, nC, nl, ltop, nfree):", [int(x) for x in q[o:o + 8]])
print({k: round(v, 3) for k, v in t.items()})
