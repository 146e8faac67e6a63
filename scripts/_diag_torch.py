import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mode = sys.argv[1]
if mode == "torch_first":
    import torch; print("torch first: avail", torch.cuda.is_available(), flush=True)
from surtr_amd import engine as E, scenes as S
eng = E.Engine(0)
print("engine up", flush=True)
if mode == "event":
    sc = S.blob_scene(64)
    eng.upload_pieces([sc["mesh"]], [sc["convex"]]); eng.upload_pattern(sc["face_off"], sc["v012"]); eng.place_cells(sc["scale"], sc["translate"])
    c = eng.fracture_event(0, 64); print("event frags", c.n_frag, flush=True)
import torch
print("device_count", torch.cuda.device_count(), flush=True)
try:
    x = torch.zeros(4096, dtype=torch.uint8, device="cuda:0"); print(mode, "OK", flush=True)
except Exception as e:
    print(mode, "FAIL", e, flush=True)
os.system("grep -E 'amdhip|hsa-runtime|rocr' /proc/%d/maps | awk '{print $6}' | sort -u" % os.getpid())
