"""Runs one case of scripts/fuzz_refracture_gpu.py as ONE event, and if the event fails saves the (piece, cell) pairs that
raised an error to gpurun_out/failing_pair_*.npz.   Usage: python scripts/find_failing_pair.py n_first n_second nu nv"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from surtr_amd import engine as E, scenes, meshgen
from oracle import oracle as O
from test_refracture import _links_symmetric
if os.environ.get("SURTR_LIB"): E._use_library_for_tests(os.path.abspath(os.environ["SURTR_LIB"]))
n_first, n_second, nu, nv = [int(x) for x in sys.argv[1:5]]
sc = scenes.make_scene(*meshgen.bumpy_torus(nu, nv), n_first)
eng = E.Engine(0)
eng.upload_pieces([sc["mesh"]], [sc["convex"]]); eng.upload_pattern(sc["face_off"], sc["v012"]); eng.place_cells(sc["scale"], sc["translate"])
eng.fracture_event(0, n_first, flags=1)
meshes, convexes = scenes.fragments_as_pieces(eng.download())
keep = [i for i, m in enumerate(meshes) if m["pos"].shape[0] >= 4 and np.diff(m["off"].astype(np.int64)).min() >= 3 and convexes[i]["pos"].shape[0] >= 4
        and _links_symmetric(m) and _links_symmetric(convexes[i])]
meshes, convexes = [meshes[i] for i in keep], [convexes[i] for i in keep]
rs = scenes.refracture_scene(meshes, convexes, n_second)
eng.upload_pieces(meshes, convexes); eng.upload_pattern(rs["face_off"], rs["v012"]); eng.place_cells_groups(rs["group_cell_off"], rs["scales"], rs["shifts"])
npairs = rs["pair_cell"].shape[0]
try:
    c = eng.fracture_pairs(rs["pair_cell"], rs["pair_piece"], flags=int(os.environ.get("SURTR_FLAGS", "3"))); print("event ok, frags", c.n_frag)
except E.SurtrError as ex:
    print("event error", ex.code)
st = eng.pair_status(npairs)
bad = np.nonzero(st)[0]
print("pairs", npairs, "with an error:", bad.tolist(), st[bad].tolist())
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
for i in bad[:4]:
    p, c = int(rs["pair_piece"][i]), int(rs["pair_cell"][i])
    f0, f1 = int(rs["face_off"][c]), int(rs["face_off"][c + 1])
    planes = O.place_cells(rs["v012"][f0:f1], rs["scales"][p], rs["shifts"][p])
    print("pair", i, "piece", p, "V", meshes[p]["pos"].shape[0], "conv V", convexes[p]["pos"].shape[0], "cell", c, "planes", f1 - f0)
    np.savez(os.path.join(ROOT, "gpurun_out", "failing_pair_%d.npz" % i), mesh_pos=meshes[p]["pos"], mesh_off=meshes[p]["off"], mesh_nbr=meshes[p]["nbr"],
             conv_pos=convexes[p]["pos"], conv_off=convexes[p]["off"], conv_nbr=convexes[p]["nbr"], planes=planes)
eng.close()
if os.environ.get("SURTR_COMPARE"):
    # compare the clip results (no render) of the whole event with the oracle, piece by piece
    from helpers import assert_event_equal
    eng = E.Engine(0)
    eng.upload_pieces(meshes, convexes); eng.upload_pattern(rs["face_off"], rs["v012"]); eng.place_cells_groups(rs["group_cell_off"], rs["scales"], rs["shifts"])
    eng.fracture_pairs(rs["pair_cell"], rs["pair_piece"], flags=1); got = eng.download(); eng.close()
    parts = []
    for p in range(len(meshes)):
        a, b = int(rs["group_cell_off"][p]), int(rs["group_cell_off"][p + 1])
        f0, f1 = int(rs["face_off"][a]), int(rs["face_off"][b])
        planes = O.place_cells(rs["v012"][f0:f1], rs["scales"][p], rs["shifts"][p])
        ev = O.event([meshes[p]], [convexes[p]], rs["face_off"][a:b + 1] - rs["face_off"][a], planes, refit=True, render=False, threads=4)
        ev["frag_ids"] = ev["frag_ids"] + np.array([a, p, 0], np.int32)
        parts.append(ev)
    ref = E.merge_fragments(parts)
    try:
        assert_event_equal(got, ref, render=False); print("clip + refit results equal the oracle's (%d fragments)" % ref["frag_ids"].shape[0])
    except AssertionError as ex:
        print("clip results DIFFER:", str(ex)[:100])
    f = 934
    a, b = int(ref["mesh_vert_off"][f]), int(ref["mesh_vert_off"][f + 1]); no = ref["mesh_nbr_off"]
    np.savez(os.path.join(ROOT, "gpurun_out", "fragment_934.npz"), pos=ref["mesh_pos"][a:b], off=(no[a:b + 1] - no[a]).astype(np.uint32), nbr=ref["mesh_nbr"][int(no[a]):int(no[b])])
    print("fragment 934: V", b - a, "H", int(no[b] - no[a]))
