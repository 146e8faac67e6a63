"""Generates the committed golden fixtures from the CPU oracle (oracle/surtr_oracle.cpp).

The reference ships no fixtures and cannot be built here (see oracle header), so these vectors are the
oracle's outputs -- itself pinned by tests/test_oracle_kat.py -- on the BASELINE.json configurations:
  cube8.npz        cfg1, complete inputs (seeds, planes) and complete outputs
  digests.json     cfg2 / cfg3 / cfg4: counts + sha256 of every output array; urchin64 / urchin1024: the deep-lobed mesh of
                   meshgen.urchin (cells with several islands, non-convex faces) at the cell counts of cfg2 / cfg3
Run from the repo root:  python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O          # noqa: E402
from surtr_amd import scenes            # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
KEYS = ("frag_ids", "mesh_vert_off", "mesh_pos", "mesh_nbr_off", "mesh_nbr", "conv_vert_off", "conv_pos", "conv_nbr_off",
        "conv_nbr", "idx_off", "idx")


def digest(ev):
    d = {k: hashlib.sha256(np.ascontiguousarray(ev[k]).tobytes()).hexdigest() for k in KEYS}
    d["n_frag"] = int(ev["frag_ids"].shape[0])
    d["mesh_verts"] = int(ev["mesh_pos"].shape[0])
    d["mesh_nbrs"] = int(ev["mesh_nbr"].shape[0])
    d["conv_verts"] = int(ev["conv_pos"].shape[0])
    d["n_idx"] = int(ev["idx"].shape[0])
    d["mesh_pos_sum"] = float(ev["mesh_pos"].astype(np.float64).sum())
    d["conv_pos_sum"] = float(ev["conv_pos"].astype(np.float64).sum())
    return d


def run(sc, threads=8):
    planes = O.place_cells(sc["v012"], sc["scale"], sc["translate"])
    return planes, O.event([sc["mesh"]], [sc["convex"]], sc["face_off"], planes, refit=True, render=True, threads=threads)


def main():
    sc = scenes.cube_scene(8)
    planes, ev = run(sc)
    np.savez_compressed(os.path.join(HERE, "cube8.npz"), seeds=sc["seeds"], face_off=sc["face_off"], v012=sc["v012"], planes=planes,
                        mesh_pos=sc["mesh"]["pos"], mesh_off=sc["mesh"]["off"], mesh_nbr=sc["mesh"]["nbr"],
                        **{"out_" + k: ev[k] for k in KEYS})
    out = {}
    for name, sc in (("blob64", scenes.blob_scene(64)), ("blob1024", scenes.blob_scene(1024)), ("torus4096", scenes.torus_scene(4096)),
                     ("urchin64", scenes.urchin_scene(64)), ("urchin1024", scenes.urchin_scene(1024))):
        _, ev = run(sc)
        out[name] = digest(ev)
        print(name, out[name]["n_frag"], out[name]["mesh_verts"], out[name]["n_idx"])
    json.dump(out, open(os.path.join(HERE, "digests.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
