import numpy as np

TOPOLOGY_KEYS = ("frag_ids", "mesh_vert_off", "mesh_nbr_off", "mesh_nbr", "conv_vert_off", "conv_nbr_off", "conv_nbr",
                 "idx_off", "idx")
COORD_KEYS = ("mesh_pos", "conv_pos", "vnc")
RTOL = 1e-5   # north_star: "float intersection coordinates within 1e-5 relative"


def assert_event_equal(got, ref, render=True):
    for k in TOPOLOGY_KEYS:
        if k in ("idx_off", "idx") and not render:
            continue
        assert got[k].shape == ref[k].shape, (k, got[k].shape, ref[k].shape)
        assert np.array_equal(got[k], ref[k]), k
    for k in COORD_KEYS:
        if k == "vnc" and not render:
            continue
        assert got[k].shape == ref[k].shape, (k, got[k].shape, ref[k].shape)
        scale = max(1.0, float(np.abs(ref[k]).max())) if ref[k].size else 1.0
        assert np.allclose(got[k], ref[k], rtol=RTOL, atol=RTOL * scale * 1e-2), k


def run_event(engine_mod, oracle, sc, flags=3, cells=None, threads=4):
    eng = engine_mod.Engine(0)
    try:
        eng.upload_pieces([sc["mesh"]], [sc["convex"]])
        eng.upload_pattern(sc["face_off"], sc["v012"])
        eng.place_cells(sc["scale"], sc["translate"])
        n = sc["n_cells"] if cells is None else cells
        c = eng.fracture_event(0, n, flags=flags)
        got = eng.download()
    finally:
        eng.close()
    planes = oracle.place_cells(sc["v012"], sc["scale"], sc["translate"])
    ref = oracle.event([sc["mesh"]], [sc["convex"]], sc["face_off"], planes, refit=bool(flags & 1), render=bool(flags & 2),
                       threads=threads, cell_end=n)
    return c, got, ref


def fragment(ev, k, which="mesh"):
    vo, no = ev[which[:4] + "_vert_off"], ev[which[:4] + "_nbr_off"]
    a, b = int(vo[k]), int(vo[k + 1])
    return {"pos": ev[which[:4] + "_pos"][a:b], "off": (no[a:b + 1] - no[a]).astype(np.uint32),
            "nbr": ev[which[:4] + "_nbr"][int(no[a]):int(no[b])]}
