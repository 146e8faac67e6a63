"""Band-size distribution of BASELINE configs[3] (cost classes of k_prep_pairs: class = 1 + n / 384 for a reduced solid of n vertices)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from surtr_amd import engine as E, scenes as S
sc = S.torus_scene(4096)
eng = E.Engine(0)
sc["convex"], _ = S.ach_convex(eng, sc["mesh"]["pos"])
eng.upload_pieces([sc["mesh"]], [sc["convex"]]); eng.upload_pattern(sc["face_off"], sc["v012"]); eng.place_cells(sc["scale"], sc["translate"])
c = eng.fracture_event(0, 4096, flags=3)
qs = eng.queue_stats()
print("fragments", c.n_frag, "clip classes (0 none, 1..12 = n/384, 13 no image, 14 big, 15 wide):", qs[16:32].tolist())
print("regular small-solid clipper (k_clip_convex): taken %d, handed to the general clipper %d" % tuple(qs[80:82].tolist()))
print("pre-pass classes:", qs[48:64].tolist(), "arena V/H/I:", qs[0:3].tolist(), "image arena 16B units:", int(qs[10]))
eng.close()
