"""Exploratory GPU-vs-oracle comparison (verbose); the real checks live in tests/."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as O
from surtr_amd import engine as E, scenes as S

def compare(got, ref, tag):
    ok = True
    for k in ("frag_ids", "mesh_vert_off", "mesh_nbr_off", "mesh_nbr", "conv_vert_off", "conv_nbr_off", "conv_nbr", "idx_off", "idx"):
        if got[k].shape != ref[k].shape or not np.array_equal(got[k], ref[k]):
            ok = False
            print("  MISMATCH", tag, k, got[k].shape, ref[k].shape)
            if got[k].shape == ref[k].shape:
                bad = np.argwhere(got[k] != ref[k])
                print("    first bad", bad[:5].tolist(), got[k].reshape(-1)[:0])
    for k in ("mesh_pos", "conv_pos", "vnc"):
        if k == "vnc" and ref[k].shape[0] == 0: continue
        if got[k].shape != ref[k].shape:
            ok = False; print("  MISMATCH shape", tag, k, got[k].shape, ref[k].shape); continue
        if got[k].size:
            d = np.abs(got[k] - ref[k]).max()
            exact = np.array_equal(got[k], ref[k])
            print("  ", tag, k, "maxabs", d, "bitexact", exact)
            if not np.allclose(got[k], ref[k], rtol=1e-5, atol=1e-6): ok = False
    return ok

def run(sc, tag, flags, cells=None):
    eng = E.Engine(0)
    eng.upload_pieces([sc["mesh"]], [sc["convex"]])
    eng.upload_pattern(sc["face_off"], sc["v012"])
    eng.place_cells(sc["scale"], sc["translate"])
    n = sc["n_cells"] if cells is None else cells
    t0 = time.time()
    c = eng.fracture_event(0, n, flags=flags)
    t1 = time.time()
    c = eng.fracture_event(0, n, flags=flags)
    t2 = time.time()
    got = eng.download()
    planes = O.place_cells(sc["v012"], sc["scale"], sc["translate"])
    ref = O.event([sc["mesh"]], [sc["convex"]], sc["face_off"], planes, refit=bool(flags & 1), render=bool(flags & 2), threads=8, cell_end=n)
    print(tag, "flags", flags, "gpu frags", c.n_frag, "ref frags", ref["frag_ids"].shape[0], "mesh verts", c.mesh_verts, ref["mesh_pos"].shape[0],
          "idx", c.n_idx, ref["idx"].shape[0], "status", c.status, "gpu ms first/second %.2f %.2f" % ((t1-t0)*1e3, (t2-t1)*1e3), "cpu s %.3f" % ref["seconds"])
    ok = compare(got, ref, tag)
    print(tag, "PARITY", "OK" if ok else "FAIL")
    if not ok:
        os.makedirs("gpurun_out", exist_ok=True)
        np.savez_compressed("gpurun_out/mismatch_%s.npz" % tag.replace("[", "_").replace("]", "").replace(":", ""),
                            **{"got_" + k: v for k, v in got.items()}, **{"ref_" + k: v for k, v in ref.items() if k != "seconds"})
    eng.close()
    return ok

if __name__ == "__main__":
    which = sys.argv[1:] or ["cube", "blob64"]
    allok = True
    for w in which:
        if w == "cube":
            sc = S.cube_scene(8)
            for fl in (0, 1, 2, 3): allok &= run(sc, "cube8", fl)
        elif w == "blob64":
            sc = S.blob_scene(64)
            for fl in (0, 3): allok &= run(sc, "blob64", fl)
        elif w == "blob1024":
            sc = S.blob_scene(1024)
            allok &= run(sc, "blob1024", 3)
        elif w.startswith("torus"):
            n = int(w[5:] or 4096)
            t0 = time.time(); sc = S.torus_scene(4096); print("scene build s", time.time() - t0)
            allok &= run(sc, "torus4096[:%d]" % n, 3, cells=n)
    print("ALL", "OK" if allok else "FAIL")
    sys.exit(0 if allok else 1)
