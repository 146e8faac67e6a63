#!/usr/bin/env python3
"""Per pair of a configs[3] event: band size, largest bucket, LDS bytes the record clipper needed at its worst plane, lane-0 cycles
(from a -DSURTR_STAMP build).  Usage: python scripts/wave_need.py build_tmp/libsurtr_hip_stamp.so"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from surtr_amd import engine as E, scenes, meshgen
E._use_library_for_tests(os.path.abspath(sys.argv[1]))
L = E.lib()
eng = E.Engine(0)
sc = scenes.mesh_scene(*meshgen.bumpy_torus(), eng=eng)
eng.build_cells(scenes.uniform_seeds(4096, scenes.SEED))
sc["convex"], _ = scenes.ach_convex(eng, sc["mesh"]["pos"])
eng.upload_pieces([sc["mesh"]], [sc["convex"]]); eng.place_cells(sc["scale"], sc["translate"])
for _ in range(2): eng.fracture_event(0, 4096)
buf = np.zeros(4 * 8192, np.uint32)
L.surtr_debug_wneed(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint32(buf.size))
d = buf.reshape(-1, 4)[:4096]
d = d[d[:, 0] > 0]
ok = d[d[:, 2] != 0xFFFFFFFF]
print("pairs through the record clipper:", len(d), "finished:", len(ok))
n, mb, need, cyc = ok[:, 0].astype(float), ok[:, 1].astype(float), ok[:, 2].astype(float), ok[:, 3].astype(float)
print("corr(need, n) %.3f  corr(need, maxbucket) %.3f  corr(cycles, n) %.3f" % (np.corrcoef(need, n)[0, 1], np.corrcoef(need, mb)[0, 1], np.corrcoef(cyc, n)[0, 1]))
A = np.stack([n, mb, np.ones_like(n)], 1); coef = np.linalg.lstsq(A, need, rcond=None)[0]; res = need - A @ coef
print("need ~ %.2f n + %.2f maxbucket + %.0f ; residual std %.0f max %.0f" % (coef[0], coef[1], coef[2], res.std(), res.max()))
for cap in (32768, 36864, 40960, 45056, 49152):
    fit = need <= cap
    print("cap %5d: %4d pairs fit (%.0f%% of pairs, %.0f%% of cycles)" % (cap, fit.sum(), 100 * fit.mean(), 100 * cyc[fit].sum() / cyc.sum()))
    # a predictor: n + 2 maxbucket below a threshold chosen so that at most 1% of the admitted pairs do not fit
    score = coef[0] * n + coef[1] * mb + coef[2]
    for margin in (0, 2048, 4096, 6144):
        adm = score + margin <= cap
        bad = (adm & ~fit).sum()
        print("      predicted + %4d <= cap: admitted %4d, of which %3d would not fit" % (margin, adm.sum(), bad))
pb = (ctypes.c_ulonglong * 32)()
L.surtr_debug_wplane(pb, 1)
eng.fracture_event(0, 4096)
L.surtr_debug_wplane(pb, 0)
tot = sum(pb[2 * c + 1] for c in range(8))
print("cutting planes by items per plane (originals the plane clips + alive cut points):")
for c in range(8):
    if pb[2 * c]:
        print("   <= %4d items: %6d planes, %6.0f cycles per plane, %4.1f%% of the plane cycles" % (32 << c, pb[2 * c], pb[2 * c + 1] / pb[2 * c], 100.0 * pb[2 * c + 1] / max(tot, 1)))
np.save("gpurun_out/wave_need.npy", d)
eng.close()
