"""GPU-vs-oracle fuzz of multi-piece events (BASELINE configs[4] shape): random first-level fracture, every fragment
re-split by its own cells in ONE event of many small pieces -- the events that take k_clip_pairs_half.
Usage: python scripts/fuzz_refracture_gpu.py [n_cases] [seed]"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import oracle as O
from surtr_amd import engine as E
from helpers import assert_event_equal, assert_event_equal_flagged
from test_refracture import _refracture, _links_symmetric
from surtr_amd import scenes as S, meshgen as G


def reference_result_is_invalid(n_first, n_second, nu, nv):
    """The engine refused the second-level event with SURTR_E_TOPOLOGY: is there a (piece, cell) whose Convex the reference
    clips into something that is not a solid (a ring entry past the last vertex, or a one-way link)?  Rebuilds the pieces
    with the engine (first level is a regular event) and clips every piece's Convex by every cell of its group."""
    sc = S.make_scene(*G.bumpy_torus(nu, nv), n_first)
    eng = E.Engine(0)
    eng.upload_pieces([sc["mesh"]], [sc["convex"]]); eng.upload_pattern(sc["face_off"], sc["v012"]); eng.place_cells(sc["scale"], sc["translate"])
    eng.fracture_event(0, n_first, flags=1)
    meshes, convexes = S.fragments_as_pieces(eng.download())
    eng.close()
    keep = [i for i, m in enumerate(meshes) if m["pos"].shape[0] >= 4 and np.diff(m["off"].astype(np.int64)).min() >= 3 and convexes[i]["pos"].shape[0] >= 4
            and _links_symmetric(m) and _links_symmetric(convexes[i])]
    meshes, convexes = [meshes[i] for i in keep], [convexes[i] for i in keep]
    rs = S.refracture_scene(meshes, convexes, n_second)
    for p in range(len(meshes)):
        a, b = int(rs["group_cell_off"][p]), int(rs["group_cell_off"][p + 1])
        for c in range(a, b):
            f0, f1 = int(rs["face_off"][c]), int(rs["face_off"][c + 1])
            planes = O.place_cells(rs["v012"][f0:f1], rs["scales"][p], rs["shifts"][p])
            for solid in (convexes[p], meshes[p]):
                O.links_off_the_array(reset=True)
                r = O.clip(solid, planes)
                if O.links_off_the_array(reset=True) > 0:      # a link that names no vertex / a clipped one: the restatement stops there
                    return True
                V = r["pos"].shape[0]
                if V == 0:
                    continue
                if int(r["nbr"].max()) >= V or int(r["nbr"].min()) < 0:
                    return True
                rings = [set(r["nbr"][r["off"][v]:r["off"][v + 1]].tolist()) for v in range(V)]
                if any(v not in rings[u] for v in range(V) for u in rings[v]):
                    return True
    # ... or a fragment on which the reference's ExtractFaces (Src/Poly.cpp:89-126, `while (cur != i)`) never ends: a walk
    # that is not back at its start vertex after H steps has repeated a (previous, current) state
    for p in range(len(meshes)):
        a, b = int(rs["group_cell_off"][p]), int(rs["group_cell_off"][p + 1])
        f0, f1 = int(rs["face_off"][a]), int(rs["face_off"][b])
        planes = O.place_cells(rs["v012"][f0:f1], rs["scales"][p], rs["shifts"][p])
        ev = O.event([meshes[p]], [convexes[p]], rs["face_off"][a:b + 1] - rs["face_off"][a], planes, refit=False, render=False, threads=4)
        vo, no, nb = ev["mesh_vert_off"], ev["mesh_nbr_off"], ev["mesh_nbr"]
        for f in range(len(vo) - 1):
            v0, v1 = int(vo[f]), int(vo[f + 1])
            ring = [nb[int(no[v]):int(no[v + 1])].tolist() for v in range(v0, v1)]
            H = sum(len(r) for r in ring)
            seen = set()
            for i in range(v1 - v0):
                for adj in ring[i]:
                    if (i, adj) in seen:
                        continue
                    prev, cur, steps = i, adj, 0
                    while cur != i:
                        seen.add((prev, cur))
                        r = ring[cur]
                        k = r.index(prev) if prev in r else len(r)
                        prev, cur = cur, (r[-1] if k == 0 or k == len(r) else r[k - 1])
                        steps += 1
                        if steps > H:
                            return True
                    seen.add((prev, cur))
    return False


def reference_result_is_invalid_child(n_first, n_second, nu, nv):
    """The same in a child process (ExtractFaces walks of the check above may be long; before round 3 the restated reference
    also read out of bounds where the reference would, and could take the process with it)."""
    rc = subprocess.call([sys.executable, os.path.abspath(__file__), "--check", str(n_first), str(n_second), str(nu), str(nv)],
                         stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return rc != 0


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--check":
        sys.exit(1 if reference_result_is_invalid(*[int(x) for x in sys.argv[2:6]]) else 0)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 99)
    bad = 0; refused = 0
    t0 = time.time()
    for case in range(n):
        n_first = int(rng.choice([6, 16, 40, 96, 200])); n_second = int(rng.choice([3, 8, 17, 32]))
        nu, nv = int(rng.integers(30, 260)), int(rng.integers(20, 200))
        try:
            c, got, ref, npieces = _refracture(E, O, n_first, n_second, nu, nv)
            assert c.status == 0
            assert_event_equal_flagged(got, ref)      # (a flagged fragment: the reference's own result for it must be invalid)
            res = "ok" if c.n_failed == 0 else "ok (%d fragment(s) flagged: no valid result in the reference)" % c.n_failed
        except AssertionError as e:
            res = "MISMATCH %s" % (e,); bad += 1
        except E.SurtrError as e:
            if e.code == E.E_TOPOLOGY and reference_result_is_invalid_child(n_first, n_second, nu, nv):
                refused += 1
                print("case %d torus %dx%d first %d second %d: refused (SURTR_E_TOPOLOGY); the reference's clip of one of the pieces is not a solid, or its ExtractFaces would not end  (%.0fs)" % (
                    case, nu, nv, n_first, n_second, time.time() - t0), flush=True)
            else:
                bad += 1
                print("case %d torus %dx%d first %d second %d: ENGINE ERROR %s" % (case, nu, nv, n_first, n_second, e), flush=True)
            continue
        print("case %d torus %dx%d first %d second %d pieces %d pairs %d frags %d %s  (%.0fs)" % (
            case, nu, nv, n_first, n_second, npieces, c.n_pairs, c.n_frag, res, time.time() - t0), flush=True)
    print("FUZZ DONE: %d cases, %d mismatches, %d inputs outside the reference's domain" % (n, bad, refused))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
