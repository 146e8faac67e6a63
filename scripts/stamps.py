import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from surtr_amd import engine as E, scenes as S
sc = S.torus_scene(4096)
if os.environ.get("SURTR_ACH"):
    _e = E.Engine(0); sc["convex"], _ = S.ach_convex(_e, sc["mesh"]["pos"]); _e.close()
lib = os.path.abspath(sys.argv[1])
E._use_library_for_tests(lib)
L = E.lib()
eng = E.Engine(0)
eng.upload_pieces([sc["mesh"]], [sc["convex"]]); eng.upload_pattern(sc["face_off"], sc["v012"]); eng.place_cells(sc["scale"], sc["translate"])
flags = int(sys.argv[2]) if len(sys.argv) > 2 else 0
c = eng.fracture_event(0, 4096, flags=flags)
buf = (ctypes.c_ulonglong * 96)()
L.surtr_debug_stamps(buf, 1)
c = eng.fracture_event(0, 4096, flags=flags)
L.surtr_debug_stamps(buf, 1)
names = ["pre: A1 stream", "pre: A2 exact", "pre: emit", "pre: A3 block counts + scan", "plane: classify", "plane: cut links patch", "plane: relink finalize/serial", "plane: tombstones",
         "plane: cut scan", "plane: cut sparse sweep", "plane: cut dense create", "plane: chain jump", "plane: walks (first steps)", "plane: pred check", "plane: resumed walks"]
tot = sum(buf[i] for i in range(15))
for i, n in enumerate(names):
    print("%-20s %14d  %5.1f%%" % (n, buf[i], 100.0 * buf[i] / max(tot, 1)))
print("total cycles (lane0, summed over WGs)", tot)
print("serial planes", buf[19], "solids redone on global scratch", buf[20], "squeezes", buf[31])
print("overflow causes: toolong %d, n>capV %d, hsum>capEmit %d, M>capAux %d, after squeeze %d, zw %d, ring len %d" % tuple(buf[32:39]))
print("regular (parallel-relink) planes: %d, with chain jumping: %d" % (buf[43], buf[40]))
print("prepass (big solids): undecided blocks %d of %d (%.1f%%), needy vertices %d" % (buf[45], buf[46], 100.0*buf[45]/max(buf[46],1), buf[47]))
print("pair cost histogram (cycles < 2^17, 2^18, ...):", [buf[21 + i] for i in range(11)])
print("per-WG busy: avg %.3g max %.3g cycles over %d WGs" % (buf[16] / max(buf[18], 1), buf[17], buf[18]))
print("big kernel: pair cost histogram:", [buf[51 + i] for i in range(10)])
print("big kernel: per-WG busy: avg %.3g max %.3g cycles over %d WGs" % (buf[48] / max(buf[50], 1), buf[49], buf[50]))
eng.close()

print("prep kernel: pair cost histogram (cycles < 2^17, 2^18, ...):", [buf[62 + i] for i in range(8)])
print("prep kernel: per-WG lifetime avg %.3g max %.3g, work avg %.3g max %.3g cycles over %d WGs" % (buf[56] / max(buf[58], 1), buf[57], buf[59] / max(buf[58], 1), buf[60], buf[58]))
print("park (lane-0 cycles): index_live %d, island labels %d, roots+arena %d, write %d; image load %d" % tuple(buf[80:85]))
print("one-wave clips: convex kernel: prepass %d, planes %d, park %d over %d tasks; refit: prepass %d, planes %d, park %d over %d tasks" % tuple(buf[86:94]))
print("refit: hull4 %d, k-DOP %d (lane-0 cycles)" % (buf[94], buf[95]))
print("select (lane-0 cycles): loop top %d, count+scan %d, write %d" % tuple(buf[76:79]))
print("prep kernel (lane-0 cycles): planes %d, select %d, image alloc %d, mask copy %d, emit %d, hist+header %d" % tuple(buf[70:76]))
if flags & 2:
    print("k_faces: successors %d, pointer jumping %d, owners+loops %d, wave ears %d, lane ears %d, compaction %d; faces > 64 vertices: %d (avg %.1f)" % (buf[60], buf[61], buf[62], buf[63], buf[64], buf[65], buf[66], buf[67] / max(buf[66], 1)))
    print("k_faces: fragment cost histogram (cycles < 2^15, 2^16, ...):", [buf[70 + i] for i in range(10)], "max", buf[68], "largest slow fragment n", buf[69])
    print("k_faces: slowest fragment: n %d H %d faces %d | succ %d jump %d own %d wave %d lane %d compact %d" % (buf[69], buf[77], buf[78], buf[32], buf[33], buf[34], buf[35], buf[36], buf[37]))
