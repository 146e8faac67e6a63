"""Per-phase cycle stamps of the one-wave clip kernels (k_clip_convex on configs[3]) from a -DSURTR_STAMP -DSURTR_STAMP_SMALL build.
Usage: python scripts/stamps_small.py build_tmp/libsurtr_stamp_small.so"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from surtr_amd import engine as E, scenes as S
sc = S.torus_scene(4096)
E._use_library_for_tests(os.path.abspath(sys.argv[1]))
L = E.lib()
eng = E.Engine(0)
sc["convex"], _ = S.ach_convex(eng, sc["mesh"]["pos"])
eng.upload_pieces([sc["mesh"]], [sc["convex"]]); eng.upload_pattern(sc["face_off"], sc["v012"]); eng.place_cells(sc["scale"], sc["translate"])
eng.fracture_event(0, 4096, flags=0)
buf = (ctypes.c_ulonglong * 96)()
L.surtr_debug_stamps(buf, 1)
eng.fracture_event(0, 4096, flags=0)
L.surtr_debug_stamps(buf, 1)
names = {1: "select: loop top / classify", 2: "select: count + scan", 3: "select: write list", 4: "flags, branch decisions", 8: "edge cuts: count + scan", 9: "edge cuts: sweep (sources)",
         10: "edge cuts: create + patch", 5: "edge cuts: tail (ordered patch)", 12: "walks (first steps)", 11: "chain jump", 14: "resumed walks", 13: "pred check", 6: "relink finalize / serial", 7: "live count / tail"}
tot = sum(buf[i] for i in names)
for i in (1, 2, 3, 4, 8, 9, 10, 5, 12, 11, 14, 13, 6, 7):
    print("%-34s %14d  %5.1f%%" % (names[i], buf[i], 100.0 * buf[i] / max(tot, 1)))
print("total plane-loop cycles (lane 0, all tasks)", tot, "; tasks", buf[89], "; planes loop per task %.0f" % (buf[87] / max(buf[89], 1)))
eng.close()
