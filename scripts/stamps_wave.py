#!/usr/bin/env python3
"""Lane-0 cycles per phase of the wave clipper (wave_clip.h) on a BASELINE configs[3] event, from a -DSURTR_STAMP build, and the
rules that made it hand pairs on.  Usage: python scripts/stamps_wave.py build_tmp/libsurtr_hip_stamp.so [cells]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from surtr_amd import engine as E, scenes, meshgen

lib = os.path.abspath(sys.argv[1])
cells = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
E._use_library_for_tests(lib)
L = E.lib()
eng = E.Engine(0)
sc = scenes.mesh_scene(*meshgen.bumpy_torus(), eng=eng)
eng.build_cells(scenes.uniform_seeds(cells, scenes.SEED))
sc["convex"], _ = scenes.ach_convex(eng, sc["mesh"]["pos"])
eng.upload_pieces([sc["mesh"]], [sc["convex"]])
eng.place_cells(sc["scale"], sc["translate"])
buf = (ctypes.c_ulonglong * 64)()
c = eng.fracture_event(0, cells)
L.surtr_debug_stamps_wave(buf, 1)
eng.set_profiling(True)
c = eng.fracture_event(0, cells)
q = eng.queue_stats()
t = eng.kernel_times()
L.surtr_debug_stamps_wave(buf, 1)
names = ["load: ranks", "load: ids + records", "plane: item scan, wait at its barrier", "plane: totals, room, carving", "plane: item scan, wait at its 2nd barrier",
         "plane: first walk steps + new vertices", "plane: pointer jumping", "plane: resumed walks", "plane: zero masks, check", "plane: bookkeeping",
         "park: lengths", "park: rings", "park: islands", "park: copy", "plane: item scan, kept masks of the items", "plane: item scan, places + tails + sources"]
order = [0, 1, 14, 2, 3, 15, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13]
tot = sum(buf[i] for i in range(len(names))) + buf[25]
for i in order:
    print("%-46s %14d  %5.1f%%" % (names[i], buf[i], 100.0 * buf[i] / max(tot, 1)))
pairs, planes = max(buf[16], 1), max(buf[18], 1)
print("pairs %d (band vertices avg %.0f), cutting planes %d (%.1f per pair): clipped avg %.1f, new avg %.1f, alive cut points avg %.1f" %
      (buf[16], buf[17] / pairs, buf[18], buf[18] / pairs, buf[19] / planes, buf[20] / planes, buf[21] / planes))
print("waiting for the slowest resumed walk (barrier): %d cycles (%.0f per cutting plane)" % (buf[25], buf[25] / planes))
print("resumed walks: steps total %d, longest %d; planes with pointer jumping %d" % (buf[22], buf[23], buf[24]))
print("cycles per pair %.0f, per cutting plane %.0f (plane phases only)" % (tot / pairs, (sum(buf[i] for i in range(2, 10)) + buf[14] + buf[15] + buf[25]) / planes))
print("LDS need of a pair at its worst plane, 4 KiB classes:", [int(buf[32 + i]) for i in range(16)])
print("  ... at its worst plane from the third on:          ", [int(buf[48 + i]) for i in range(16)])
print("fragments", c.n_frag, "| wave took", q[88], "handed on", q[89], "| by rule:", {i: q[96 + i] for i in range(1, 20) if q[96 + i]})
print({k: round(v, 3) for k, v in t.items()})
eng.close()
