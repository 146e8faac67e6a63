"""Replays one case of scripts/fuzz_refracture_gpu.py piece by piece (engine vs oracle) and saves the pieces that differ.
Usage: python scripts/repro_refracture_case.py n_first n_second nu nv [first_piece]   (SURTR_LIB=tests/emul/libsurtr_emul.so for the CPU emulation)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from surtr_amd import engine as E, scenes, meshgen
from oracle import oracle as O
from helpers import assert_event_equal
from test_refracture import _links_symmetric
if os.environ.get('SURTR_LIB'): E._use_library_for_tests(os.path.abspath(os.environ['SURTR_LIB']))
n_first, n_second, nu, nv = [int(x) for x in sys.argv[1:5]]
sc = scenes.make_scene(*meshgen.bumpy_torus(nu, nv), n_first)
eng = E.Engine(0)
eng.upload_pieces([sc["mesh"]], [sc["convex"]]); eng.upload_pattern(sc["face_off"], sc["v012"]); eng.place_cells(sc["scale"], sc["translate"])
eng.fracture_event(0, n_first, flags=1)
first = eng.download(); eng.close()
meshes, convexes = scenes.fragments_as_pieces(first)
keep = [i for i, m in enumerate(meshes) if m["pos"].shape[0] >= 4 and np.diff(m["off"].astype(np.int64)).min() >= 3 and convexes[i]["pos"].shape[0] >= 4 and _links_symmetric(m) and _links_symmetric(convexes[i])]
meshes, convexes = [meshes[i] for i in keep], [convexes[i] for i in keep]
rs = scenes.refracture_scene(meshes, convexes, n_second)
print("pieces", len(meshes), "pairs", rs["pair_cell"].shape[0], flush=True)
e2 = E.Engine(0)
log = open(os.path.join(ROOT, "gpurun_out", "repro_refracture.log"), "w") if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else sys.stdout
for p in range(int(sys.argv[5]) if len(sys.argv) > 5 else 0, len(meshes)):
    if p % 20 == 0: print("piece", p, file=log, flush=True)
    a, b = int(rs["group_cell_off"][p]), int(rs["group_cell_off"][p + 1])
    f0, f1 = int(rs["face_off"][a]), int(rs["face_off"][b])
    planes = O.place_cells(rs["v012"][f0:f1], rs["scales"][p], rs["shifts"][p])
    fo = (rs["face_off"][a:b + 1] - rs["face_off"][a]).astype(np.uint32)
    e2.upload_pieces([meshes[p]], [convexes[p]]); e2.upload_planes(fo, planes)
    st = "ok"
    try:
        cnt = e2.fracture_event(0, b - a, flags=3); got = e2.download()
    except E.SurtrError as ex:
        st = "engine error %d" % ex.code
    ev = O.event([meshes[p]], [convexes[p]], fo, planes, refit=True, render=True, threads=2)
    if st == "ok":
        try:
            assert_event_equal(got, ev)
        except AssertionError as ex:
            st = "MISMATCH %s" % str(ex)[:60]
    if st != "ok":
        print("piece", p, "V", meshes[p]["pos"].shape[0], st, "| oracle frags", ev["frag_ids"].shape[0], flush=True)
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True); np.savez(os.path.join(ROOT, "gpurun_out", "refracture_piece_%d.npz" % p), mesh_pos=meshes[p]["pos"], mesh_off=meshes[p]["off"], mesh_nbr=meshes[p]["nbr"], conv_pos=convexes[p]["pos"], conv_off=convexes[p]["off"], conv_nbr=convexes[p]["nbr"], planes=planes, fo=fo)
print("done")
