"""Where the slowest pair of k_clip_convex spends its cycles (-DSURTR_STAMP build).  Usage: python scripts/stamps_convex.py build_tmp/libsurtr_hip_stamp.so"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from surtr_amd import engine as E, scenes as S
E._use_library_for_tests(os.path.abspath(sys.argv[1]))
L = E.lib()
eng = E.Engine(0)
for name, sc, n in (("blob64", S.blob_scene(64), 64), ("blob1024", S.blob_scene(1024), 1024), ("torus4096", S.torus_scene(4096), 4096)):
    sc["convex"], _ = S.ach_convex(eng, sc["mesh"]["pos"])
    eng.upload_pieces([sc["mesh"]], [sc["convex"]]); eng.upload_pattern(sc["face_off"], sc["v012"]); eng.place_cells(sc["scale"], sc["translate"])
    eng.fracture_event(0, n)
    buf = (ctypes.c_ulonglong * 64)()
    L.surtr_debug_stamps2(buf, 1)
    eng.fracture_event(0, n)
    L.surtr_debug_stamps2(buf, 1)
    print("%s: pairs %d, mean %d cycles, slowest %d: planes+small_clip %d, park %d, rest (estimate, record) %d; F %d" % (name, buf[33], buf[32] // max(buf[33], 1), buf[34], buf[35], buf[36], buf[37], buf[38]))
    print("   small_clip of that pair: load %d, classify %d, numbering %d, relink %d, check %d, compaction %d; cutting planes %d" % tuple(buf[40 + q] for q in range(7)))
eng.close()
