"""Multi-GPU plumbing: one process per GPU, cells sharded in contiguous blocks, one all-gather of the packed
fragment blobs per event (RCCL over xGMI with backend "nccl"; gloo on CPU in the tests).

torch is used only for device memory and torch.distributed.
"""
import numpy as np
import torch
import torch.distributed as dist

from . import engine


def pack_blob(eng, counts, device):
    """Packs the last event of `eng` into a uint8 tensor on `device` (the engine writes it through the C ABI)."""
    nbytes = engine.blob_bytes(counts)
    blob = torch.empty(nbytes, dtype=torch.uint8, device=device)
    eng.pack_dev(blob.data_ptr(), nbytes)
    return blob


def allgather_blobs(eng, counts, device):
    """Returns the list of per-rank blobs (uint8 tensors), rank order = cell order."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    blob = pack_blob(eng, counts, device)
    if world == 1:
        return [blob]
    sizes = torch.zeros(world, dtype=torch.int64, device=device)
    mine = torch.tensor([blob.numel()], dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(sizes, mine)
    sizes_h = sizes.cpu().tolist()
    cap = max(sizes_h)
    padded = torch.zeros(cap, dtype=torch.uint8, device=device)
    padded[:blob.numel()] = blob
    gathered = torch.empty(world * cap, dtype=torch.uint8, device=device)
    dist.all_gather_into_tensor(gathered, padded)
    return [gathered[r * cap:r * cap + sizes_h[r]] for r in range(world)]


def allgather_fragments(eng, counts, device):
    """All-gather + host-side merge into one cell-major fragment set (used by tests and the harness)."""
    blobs = allgather_blobs(eng, counts, device)
    if isinstance(device, str) and device == "cpu" or getattr(device, "type", None) == "cpu":
        pass
    else:
        torch.cuda.synchronize()
    parts = []
    for b in blobs:
        _, fr = engine.unpack_blob(b.cpu().numpy())
        parts.append(fr)
    return engine.merge_fragments(parts)
