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
b2z = (ctypes.c_ulonglong * 64)(); L.surtr_debug_stamps2(b2z, 1)
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
b2 = (ctypes.c_ulonglong * 64)()
L.surtr_debug_stamps2(b2, 0)
print("k_refit: task cost histogram (cycles < 2^13, 2^14, ...):", [b2[i] for i in range(16)])
print("k_refit: slowest task %d cycles (mesh n %d, convex n %d); sum %d; WG lifetime avg %.3g max %.3g over %d WGs, most tasks in one WG %d" % (b2[16], b2[17], b2[18], b2[19], b2[20] / max(b2[22], 1), b2[21], b2[22], b2[23]))
print("k_faces: task cost histogram (cycles < 2^13, 2^14, ...):", [b2[32 + i] for i in range(16)])
print("k_faces: slowest task %d cycles (n %d H %d faces %d | succ %d jump %d own %d wave %d lane %d compact %d); sum %d; WG lifetime avg %.3g max %.3g over %d WGs" % (b2[48], b2[49], b2[50], b2[55], b2[56], b2[57], b2[58], b2[59], b2[60], b2[61], b2[51], b2[52] / max(b2[54], 1), b2[53], b2[54]))
