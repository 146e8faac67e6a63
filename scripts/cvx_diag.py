import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from surtr_amd import engine as E, scenes as S
import numpy as np
eng = E.Engine(0)
for name, sc, n in (("blob64", S.blob_scene(64), 64), ("blob1024", S.blob_scene(1024), 1024), ("torus4096", S.torus_scene(4096), 4096)):
    sc["convex"], _ = S.ach_convex(eng, sc["mesh"]["pos"])
    eng.upload_pieces([sc["mesh"]], [sc["convex"]]); eng.upload_pattern(sc["face_off"], sc["v012"]); eng.place_cells(sc["scale"], sc["translate"])
    c = eng.fracture_event(0, n)
    qs = eng.queue_stats()
    fo = np.asarray(sc["face_off"]); 
    print(name, "convex verts", len(sc["convex"]["pos"]) if isinstance(sc["convex"], dict) else "?", "small clip took", qs[80], "fell back", qs[81], "| refit: took", qs[82], "fell back", qs[83], "| general clips resumed from a later plane: convex", qs[78], "refit", qs[79], "faces per cell: mean %.1f max %d" % (np.diff(fo).mean(), np.diff(fo).max()))
eng.close()
