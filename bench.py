#!/usr/bin/env python3
"""bench.py -- fragments/sec + ms/fracture-event on BASELINE.json's configs[3]:
100k-triangle synthetic mesh (bumpy torus, 50 000 v / 100 000 tri) x 4096 Voronoi cells.

A "step" is one fracture event per GPU on device-resident inputs (by default three independent events are in flight per GPU --
three engines on three streams take the steps in turn, so that the under-filled tails of one event's kernels run beside the next
event's; `--in-flight 1` runs them one after the other, and the JSON line carries that single-event latency too as
`ms_per_fracture_event`): cell placement (A3), clip of Convex and
Mesh against every cell + island split (A7, A8, A11), refit (A12), face extraction + ear clipping
(A9, A10), pack of the fragment blob, and -- for N > 1 -- one all-gather of the blobs over RCCL, issued on
RCCL's stream so that it runs beside the kernels of the next event (two blob buffers); the timed region ends
with every gather complete.

Scaling: BASELINE's metric is the ONE 4096-cell event of configs[3] at 1 / 2 / 4 / 8 GPUs, so the headline at N > 1
is that event with its cells sharded over the ranks in contiguous blocks ("strong": total work fixed; no data-path
collective but the final all-gather of fragment blobs).  The path also partitions into independent (cell, piece)
units; the same run times the weak-scaled job too -- every rank one 4096-cell pattern of its own (seed 46354 + rank)
-- and reports it under the extra key `weak_scaled` (`--scaling weak` makes that the headline instead).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`--gpus N` MEANS N ranks: started without a launcher (no WORLD_SIZE in the environment) and with N > 1, bench.py starts the N
rank processes itself -- fresh children of `python -m torch.distributed.run`, before anything in this process touches the GPU --
and exits with their code; with fewer than N devices visible it exits non-zero with a message instead of printing an N = 1 line.
Under a launcher WORLD_SIZE must equal N.  The printed line always has n_gpus == N.  (This is the fan-out of
Src/Surtr.cpp:2129-2146 -- one pool task per cell -- as one rank per GPU.)

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel (the Mesh clip: k_clip_pairs_wave, or k_clip_pairs when the
engine leaves the record clipper off), timed live with
HIP events on the work stream; `cpu_baseline` is the CPU oracle ("port") run on the host cores on
rank 0 at N=1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
IN_FLIGHT_DEFAULT = 6      # events in flight per GPU, at every N
# The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), and streams that share a
# queue run one behind the other.  E events in flight are E contexts x 3 streams: with four queues the kernels of different events
# -- and the side-by-side kernels of ONE event -- wait for each other although nothing orders them.  One queue per stream
# (measured on MI355X, configs[3], 60 steps: 4 queues 2.31-2.35 ms per step, 12: 2.17, 16: 2.15-2.18, 32: 2.11-2.14, 64: 2.14-2.16;
# 512-cell blocks with three in flight 0.81 -> 0.69 ms).  A setting of this process's HIP runtime, made before it starts; whoever
# embeds the engine with several contexts wants the same.  An explicit value in the environment wins.
# (events in flight, with 32 queues and the contexts told how many they are: 3 / 4 / 5 / 6 / 8 -> 2.12 / 2.08 / 2.09 / 2.065 / 2.76 ms per step)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
# CPU tier of the tests only: the single-lane emulation of the kernels (tests/emul) + gloo instead of the HIP library + RCCL, so that
# the launcher and the N-rank code path can be driven without a GPU.  Never set by the driver; the line then says "data": "emulation".
EMUL_LIB = os.environ.get("SURTR_BENCH_EMUL_LIB")


def launch_ranks(n):
    """`--gpus n` without a launcher: n fresh rank processes (torch.distributed.run), started before this process has touched the
    GPU; their stdout (rank 0's ONE JSON line) is ours.  Returns the exit code."""
    import socket
    import subprocess
    if not EMUL_LIB:
        import torch
        have = torch.cuda.device_count()          # (counting devices does not initialise the GPU)
        if have < n:
            sys.stderr.write("bench.py: --gpus %d asked for, %d device(s) visible: not running (an n_gpus=%d line would be a lie)\n" % (n, have, have))
            return 2
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def kernel_build_id():
    """Hash of the sources of the clip kernels: ties profiles/traffic.json to the build it was measured on."""
    import hashlib
    h = hashlib.sha256()
    for f in ("surtr_hip.hip", "clip_core.h", "surtr_ctx.h", "small_clip.h", "literal_clip.h", "wave_clip.h", "prep_sorted.h"):
        h.update(open(os.path.join(ROOT, "surtr_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def algorithmic_bytes(sc, counts, n_faces):
    """compulsory traffic of one event, from the run's actual counts."""
    V = sc["mesh"]["pos"].shape[0]
    H = sc["mesh"]["nbr"].shape[0]
    cV = sc["convex"]["pos"].shape[0]
    cH = sc["convex"]["nbr"].shape[0]
    C = sc["n_cells"]
    b_in = (12 * V + 4 * (V + 1) + 4 * H) + (12 * cV + 4 * (cV + 1) + 4 * cH) + 16 * n_faces + 4 * (C + 1)
    nf = counts.n_frag
    solids = (12 * counts.mesh_verts + 4 * (counts.mesh_verts + nf) + 4 * counts.mesh_nbrs) + \
             (12 * counts.conv_verts + 4 * (counts.conv_verts + nf) + 4 * counts.conv_nbrs)
    render = 36 * counts.mesh_verts + 4 * counts.n_idx
    ids = 12 * nf
    # per kernel (round 4: bytes are charged to the kernel that moves them): the Mesh clip reads the Mesh CSR of the pieces and
    # the planes and writes the Mesh solids + ids; the Convex CSR in and out belongs to k_clip_convex / k_refit
    mesh_in = (12 * V + 4 * (V + 1) + 4 * H) + 16 * n_faces + 4 * (C + 1)
    mesh_out = 12 * counts.mesh_verts + 4 * (counts.mesh_verts + nf) + 4 * counts.mesh_nbrs
    return {"in": b_in, "solids": solids, "render": render, "ids": ids, "event": b_in + solids + render + ids,
            "clip_kernel": mesh_in + mesh_out + ids}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--cells", type=int, default=4096)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--settle", type=int, default=100, help="events run during set-up, before the warm-up steps")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="strong")
    ap.add_argument("--in-flight", type=int, default=IN_FLIGHT_DEFAULT,
                    help="independent events in flight per GPU: E engines (contexts) on E streams take the steps in turn, so that the "
                         "under-filled tail of one event's kernels runs beside the next event's.  The SAME default at every N (a 1 -> 8 "
                         "curve must not mix settings); the single-event figures are reported beside it (`single_event`)")
    ap.add_argument("--balanced-blocks", action="store_true", help="strong sharding in cost-balanced contiguous cell blocks (every rank runs the whole "
                                                                   "event once at set-up to get the costs) instead of equal-sized ones; measured worthless in "
                                                                   "round 3 (profiles/r03_slices.txt), kept as an experiment")
    ap.add_argument("--equal-blocks", action="store_true", help="(the default; accepted for compatibility)")
    ap.add_argument("--torus", type=int, nargs=2, default=None, metavar=("NU", "NV"), help="grid of the bumpy torus (default 250 200 = BASELINE configs[3]); tests use a small one")
    ap.add_argument("--force-dist", action="store_true", help="rehearsal: take the N > 1 code path (process group, all-gather) with one rank")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %s rank(s)" % (args.gpus, os.environ.get("WORLD_SIZE")))
    # stdout carries the ONE JSON line and nothing else: libraries that print there (RCCL's version banner under
    # NCCL_DEBUG=VERSION) are sent to stderr
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import numpy as np
    import torch
    import torch.distributed as dist

    if args.in_flight <= 0:
        args.in_flight = IN_FLIGHT_DEFAULT
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    emul = bool(EMUL_LIB)
    if emul:
        import contextlib
        dev = torch.device("cpu")
        dev_index = 0

        class _NoStream:
            cuda_stream = 0
        new_stream = current_stream = _NoStream
        on_stream = lambda st: contextlib.nullcontext()
        dev_sync = lambda: None
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a HIP device: the engine has no CPU fallback")
        if torch.cuda.device_count() <= local_rank:
            raise SystemExit("bench.py: rank %d has no device (%d visible)" % (local_rank, torch.cuda.device_count()))
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        dev_index = local_rank
        new_stream, current_stream, on_stream, dev_sync = torch.cuda.Stream, torch.cuda.current_stream, torch.cuda.stream, torch.cuda.synchronize
    multi = world > 1 or args.force_dist
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        if emul:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    # build-if-missing: rank 0 decides for everybody (a per-rank check would let a late rank skip the barrier)
    need = torch.tensor([0 if (emul or os.path.exists(os.path.join(ROOT, "surtr_amd", "libsurtr_hip.so"))) else 1], dtype=torch.int32, device=dev)
    if multi:
        dist.broadcast(need, src=0)
    if int(need.item()):
        if rank == 0:
            import __graft_entry__
            __graft_entry__.build()        # writes a temporary file and renames it: no rank can load a half-written library
        if multi:
            dist.barrier()
    from surtr_amd import engine, scenes, multigpu
    if emul:
        engine._use_library_for_tests(EMUL_LIB)

    flags = engine.EVT_REFIT | engine.EVT_RENDER
    from surtr_amd import meshgen
    boot = engine.Engine(dev_index)
    torus = meshgen.bumpy_torus(*args.torus) if args.torus else meshgen.bumpy_torus()
    base = scenes.mesh_scene(*torus, eng=boot)      # BASELINE configs[3]'s piece; rings built on the device
    base["n_cells"] = args.cells
    boot.close()

    made = {}

    def setup(mode):
        """Engines + inputs of this rank for one scaling mode; returns the step closure and what it gathers into.
        (The second mode of a run takes the contexts and streams of the first: new ones would land on hardware queues that the
        first set's streams still occupy, and the extra figure would measure that.)"""
        E = max(1, args.in_flight)
        engs, streams = made.get("engs", []), made.get("streams", [])
        for k in range(len(engs), E):
            st = current_stream() if E == 1 else new_stream()
            e = engine.Engine(dev_index)
            if not emul:
                e.set_stream(st.cuda_stream)
            e.set_events_in_flight(E)      # (E contexts busy on this GPU: events of few pairs -- a rank's block -- take the lean kernels)
            engs.append(e); streams.append(st)
        made["engs"], made["streams"] = engs, streams
        eng = engs[0]
        sc = dict(base)
        # the Voronoi pattern is built on the device (surtr_build_cells) and stays there as every engine's pattern
        seeds = scenes.uniform_seeds(args.cells, scenes.SEED + (rank if mode == "weak" else 0))
        t0 = time.perf_counter()
        eng.build_cells(seeds)
        setup_ms["build_cells_ms"] = (time.perf_counter() - t0) * 1e3
        cells = eng.download_cells()
        sc["seeds"], sc["face_off"], sc["v012"] = seeds, cells["cell_face_off"], cells["v012"]
        # the piece's Convex is the reference's ACH (PrepareFracture steps 1-6), built once at set-up
        sc["convex"], _ = scenes.ach_convex(eng, sc["mesh"]["pos"])
        for k, e in enumerate(engs):
            if k:
                e.build_cells(seeds)
            e.upload_pieces([sc["mesh"]], [sc["convex"]])
            e.upload_pieces([sc["mesh"]], [sc["convex"]])       # second call: steady state of the piece pool
        setup_ms["upload_ms"], setup_ms["upload_allocs"] = eng.upload_stats()
        if mode == "weak" or world == 1:
            cb, ce = 0, sc["n_cells"]
        elif not args.balanced_blocks:
            cb, ce = engine.cell_block(rank, world, sc["n_cells"])
        else:
            # strong sharding in contiguous, cost-balanced cell blocks: every rank runs the whole event once at set-up (inputs
            # are replicated, so all ranks see the same per-cell costs and compute the same cuts); order is untouched
            eng.place_cells(sc["scale"], sc["translate"])
            eng.fracture_event(0, sc["n_cells"], flags=flags)
            costs = eng.pair_costs(sc["n_cells"]).astype(np.float64)      # one piece: pair = cell
            cuts = engine.balanced_blocks(costs, world)
            cb, ce = cuts[rank], cuts[rank + 1]
            setup_ms["block_cost_share"] = float(costs[cb:ce].sum() / max(costs.sum(), 1.0))
        # sizing pass: blob capacity for the timed loop (sizes are identical every step: same inputs)
        counts = None
        for e in engs:
            e.place_cells(sc["scale"], sc["translate"])
            counts = e.fracture_event(cb, ce, flags=flags)
        # (set-up, not warm-up steps: the event a few dozen times more, so that the GPU's clocks and the pools' allocations are
        #  those of a running application when the W warm-up steps and the K timed steps start -- the timed region is ~50 ms)
        for r in range(args.settle):
            engs[r % len(engs)].fracture_event_async(cb, ce, flags=flags)
        dev_sync()
        if not emul:
            free_b, total_b = torch.cuda.mem_get_info(dev)
            setup_ms["hbm_used_gb"] = round((total_b - free_b) / 2.0 ** 30, 1)      # (all the contexts of this rank, pools included)
        cap_t = torch.tensor([engine.blob_bytes(counts)], dtype=torch.int64, device=dev)
        if multi:
            dist.all_reduce(cap_t, op=dist.ReduceOp.MAX)
        cap = (int(cap_t.item()) + 4095) // 4096 * 4096
        # one blob / gather buffer per slot.  N > 1: the all-gather of event i (RCCL's stream) runs beside the kernels of event
        # i+1; the event that reuses a slot waits for that slot's gather before it packs into the same blob.  At least two slots.
        nslots = max(E, 2 if multi else 1)
        bufs = []
        for k in range(nslots):
            b = torch.zeros(cap, dtype=torch.uint8, device=dev)
            bufs.append((b, torch.zeros(world * cap, dtype=torch.uint8, device=dev) if multi else b))
        while len(pending) < nslots:
            pending.append(None)
        state = {"i": 0}

        def step():
            k = state["i"] % nslots
            e, st = engs[state["i"] % E], streams[state["i"] % E]
            state["i"] += 1
            b, g = bufs[k]
            if pending[k] is not None:
                # Work.wait() orders the CURRENT stream behind the collective: it must be the stream of the engine that packs
                # into this slot next, or the pack below could overwrite a blob the gather is still reading
                with on_stream(st):
                    pending[k][0].wait()
                pending[k] = None
            e.place_cells(sc["scale"], sc["translate"])
            e.fracture_event_async(cb, ce, flags=flags)
            e.pack_dev(b.data_ptr(), cap)
            if multi:
                with on_stream(st):
                    pending[k] = (dist.all_gather_into_tensor(g, b, async_op=True), st)

        def last():
            """(blob, gathered) of the most recent step."""
            return bufs[(state["i"] - 1) % nslots]
        return sc, engs, step, last, cap, (cb, ce)

    pending = [None, None]
    setup_ms = {}

    def fence():
        for k in range(len(pending)):
            if pending[k] is not None:
                with on_stream(pending[k][1]):
                    pending[k][0].wait()
                pending[k] = None
        if multi:
            dist.barrier()
        dev_sync()

    def timed(step, warmup, steps):
        for _ in range(warmup):
            step()
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        fence()
        dt_t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        if multi:
            dist.all_reduce(dt_t, op=dist.ReduceOp.MAX)
        return float(dt_t.item())

    sc, engs, step, last, cap, (cb, ce) = setup(args.scaling)
    eng = engs[0]
    dt = timed(step, args.warmup, args.steps)
    blob, gathered = last()

    # fragment totals over ranks (from the gathered blob headers)
    host = gathered.cpu().numpy()
    total_frag = 0
    parts = []
    for r in range(world):
        c, fr = engine.unpack_blob(host[r * cap:(r + 1) * cap])
        assert c.status == 0, "device status %d on rank %d" % (c.status, r)
        total_frag += c.n_frag
        parts.append((c, fr))

    # latency of ONE event with nothing else on the GPU (engine 0 alone, same work): what `ms_per_step` would be with --in-flight 1
    lat = []
    for _ in range(max(3, min(args.steps, 10))):
        fence()
        t0 = time.perf_counter()
        eng.place_cells(sc["scale"], sc["translate"])
        eng.fracture_event_async(cb, ce, flags=flags)
        eng.pack_dev(blob.data_ptr(), cap)
        dev_sync()
        lat.append((time.perf_counter() - t0) * 1e3)
    single_event_ms = float(np.median(lat))

    # kernel timing with HIP events on the streams the kernels run on.  The dominant kernel's launch duration is taken UNDER THE
    # CONDITIONS OF THE TIMED LOOP: the same step closure, the same number of events in flight, no synchronisation between the
    # events (surtr_kernel_history), the first and the last round of engines left out (the pipeline fills / drains there).
    # The per-kernel table (`kernel_ms`) is of one event alone on the GPU.
    E = len(engs)
    clip_ms, all_ms = [], {}
    if not emul:
        for e in engs:
            e.set_profiling(True)
        rounds = 12
        for _ in range(rounds * E):
            step()
        fence()
        for e in engs:
            h = [t for t in e.kernel_history() if t > 0]
            clip_ms += h[1:-1] if len(h) > 2 else h
        for _ in range(3):
            eng.place_cells(sc["scale"], sc["translate"])
            eng.fracture_event_async(cb, ce, flags=flags)
            eng.pack_dev(blob.data_ptr(), cap)
            for k, v in eng.kernel_times().items():
                all_ms.setdefault(k, []).append(v)
        for e in engs:
            e.set_profiling(False)
    if not clip_ms:
        clip_ms = [float("nan")]

    other_extra = None
    if multi:
        # the same run also times the other scaling mode (extra key, never the headline)
        other = "weak" if args.scaling == "strong" else "strong"
        fence()
        sc2, engs2, step2, last2, cap2, _ = setup(other)
        k2 = max(3, min(args.steps, 10))
        dt2 = timed(step2, 2, k2)
        host2 = last2()[1].cpu().numpy()
        nf2 = sum(engine.unpack_blob(host2[r * cap2:(r + 1) * cap2])[0].n_frag for r in range(world))
        other_extra = {"scaling": other, "ms_per_step": dt2 / k2 * 1e3, "fragments": nf2, "fragments_per_s": nf2 / (dt2 / k2), "steps": k2,
                       "cells": args.cells * (world if other == "weak" else 1)}
        for e in engs2:
            e.close()
        engs = [engine.Engine(dev_index)]       # (closed below)
        eng = engs[0]

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = total_frag / (dt / args.steps)
        counts0 = parts[0][0]
        n_faces_rank = int(sc["face_off"][ce] - sc["face_off"][cb])
        sc_rank = dict(sc, n_cells=ce - cb)
        ab = algorithmic_bytes(sc_rank, counts0, n_faces_rank)
        clip_avg_ms = float(np.mean(clip_ms))
        # the Mesh clip runs in k_clip_pairs_wave (record clipper) or k_clip_pairs (general clipper), whichever the engine chose
        dom = "k_clip_pairs_wave" if np.mean(all_ms.get("clip_pairs_wave", [0.0])) > np.mean(all_ms.get("clip_pairs", [0.0])) else "k_clip_pairs"
        dom_slot = dom[2:]
        if dom == "k_clip_pairs_wave" and np.mean(all_ms.get("clip_pairs_catch", [0.0])) > 0.02:
            dom = "k_clip_pairs_main"       # split arrangement (round 4): the record clipper alone; same timing slot as k_clip_pairs_wave
        achieved = ab["clip_kernel"] / (clip_avg_ms * 1e-3) / 1e9
        # HBM traffic of the kernel from the rocprofv3 PMC passes (scripts/pmc.sh), only while it describes THIS build:
        # profiles/traffic.json records the hash of the kernel sources it was measured on
        traffic, traffic_prep, traffic_note = None, None, "no profiles/traffic.json"
        tj = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tj):
            try:
                rec = json.load(open(tj))
                if rec.get("build_id") == kernel_build_id() and rec.get("kernel", "k_clip_pairs") == dom:
                    traffic, traffic_note = rec.get("k_clip_pairs_hbm_bytes_per_launch"), "rocprofv3 PMC, build %s" % rec.get("build_id")
                    traffic_prep = rec.get("k_prep_pairs_hbm_bytes_per_launch")
                else:
                    traffic_note = "profiles/traffic.json was measured on build %s, this is %s: not reported" % (rec.get("build_id"), kernel_build_id())
            except Exception as ex:
                traffic_note = "unreadable: %r" % (ex,)
        assert world == args.gpus, (world, args.gpus)
        prep_ms = float(np.mean(all_ms.get("prep_pairs", [float("nan")])))
        dom_alone_ms = float(np.mean(all_ms.get(dom_slot, [float("nan")])))
        out = {
            "metric": "fragments/sec", "value": value, "unit": "fragments/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f32", "data": "emulation (CPU tier of the tests: no GPU, not a measurement)" if emul else "synthetic",
            "config": {"workload": "bumpy torus %d v / %d tri x %d Voronoi cells (%s), "
                                   "1 piece (mesh + its ACH convex), refit + triangulation on" % (
                                       sc["mesh"]["pos"].shape[0], sc["mesh"]["nbr"].shape[0] // 3, sc["n_cells"],
                                       "BASELINE configs[3]" if (not args.torus and args.cells == 4096) else "NOT the BASELINE configuration"),
                       "cells": sc["n_cells"] * (world if args.scaling == "weak" else 1), "fragments": total_frag,
                       "events_in_flight": max(1, args.in_flight), "setup_events": args.settle,
                       "hip_hw_queues": int(os.environ.get("GPU_MAX_HW_QUEUES", "4")),
                       "parallelism": ("%d ranks x one %d-cell event each, one all-gather" % (world, sc["n_cells"])) if args.scaling == "weak"
                                      else ("one event, cells sharded x%d in contiguous %s blocks, one all-gather" % (world, "cost-balanced" if args.balanced_blocks else "equal-sized") if world > 1 else "one GPU")},
            "ms_per_fracture_event": single_event_ms,
            # the figure comparable with the reference, the cpu_baseline and rounds 1-2: ONE event at a time (this rank's block of it at N > 1)
            "single_event": {"ms": single_event_ms, "fragments_per_s": parts[0][0].n_frag / (single_event_ms * 1e-3), "fragments": parts[0][0].n_frag,
                             "note": "engine 0 alone on the GPU, place + event + pack, median of %d" % len(lat)},
            "kernel_ms": {k: float(np.mean(v)) for k, v in all_ms.items()},
            "roofline": {"kernel": dom, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_note,
                         "algorithmic_bytes_per_launch": ab["clip_kernel"], "event_algorithmic_bytes": ab["event"],
                         "avg_launch_ms": clip_avg_ms,
                         # the same kernel with nothing else on the GPU (one event at a time): with several events in flight a
                         # launch shares the CUs with the kernels of the other events and its duration depends on how the
                         # streams happen to interleave
                         "launch_ms_min_max": [float(np.min(clip_ms)), float(np.max(clip_ms))], "launches_timed": len(clip_ms),
                         "avg_launch_ms_alone": dom_alone_ms,
                         # one launch per step: average duration / time per step = how many launches of this kernel run at once
                         # (each hardware queue its own stream: the launches of the events in flight overlap, each on part of
                         # the CUs, and `achieved` -- bytes of ONE launch over ITS duration -- halves when two share the GPU);
                         # all the launches running at a time together move `achieved_all_launches`
                         "launches_in_flight": clip_avg_ms / ms_per_step,
                         "achieved_all_launches": achieved * max(1.0, clip_avg_ms / ms_per_step),
                         "frac_all_launches": achieved * max(1.0, clip_avg_ms / ms_per_step) / HBM_PEAK_GBS},
            # pre-pass + Mesh clip TOGETHER (one event alone on the GPU): the band of every pair is what the path moves between the
            # two, so their summed counter traffic against the same algorithmic bytes is the honest re-read factor
            "roofline_front_half": {"kernels": [rec.get("prep_kernel", "k_prep_pairs") if traffic_prep is not None else "k_prep_pairs", dom], "bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                                    "algorithmic_bytes": ab["clip_kernel"], "ms": prep_ms + dom_alone_ms,
                                    "achieved": ab["clip_kernel"] / ((prep_ms + dom_alone_ms) * 1e-3) / 1e9,
                                    "frac": ab["clip_kernel"] / ((prep_ms + dom_alone_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                    "traffic": (traffic + traffic_prep) if (traffic is not None and traffic_prep is not None) else None,
                                    "traffic_over_algorithmic": ((traffic + traffic_prep) / ab["clip_kernel"]) if (traffic is not None and traffic_prep is not None) else None},
        }
        out["setup"] = {k: (round(v, 3) if isinstance(v, float) else v) for k, v in setup_ms.items()}
        if other_extra is not None:
            out["weak_scaled" if other_extra["scaling"] == "weak" else "strong_sharded"] = other_extra
        if world == 1 and not args.no_cpu_baseline:
            from oracle import oracle
            threads = 16          # the reference's pool: dp::thread_pool g_threadPool(16), Src/Surtr.cpp:28
            planes = oracle.place_cells(sc["v012"], sc["scale"], sc["translate"])
            ref = oracle.event([sc["mesh"]], [sc["convex"]], sc["face_off"], planes, refit=True, render=True, threads=threads)
            nref = int(ref["frag_ids"].shape[0])
            out["cpu_baseline"] = {"value": nref / ref["seconds"], "unit": "fragments/s", "cores": threads, "kind": "port",
                                   "sample": "whole event, %d cells, one task per cell on %d worker threads, %.2f s wall "
                                             "(host has %d logical cores)" % (sc["n_cells"], threads, ref["seconds"], os.cpu_count()),
                                   "ms_per_event": ref["seconds"] * 1e3}
            # SURVEY 8(d) also asks for 1 thread and all host cores: 1 thread on every 8th 64-cell block (bounded), all cores whole
            t1_frag, t1_sec = 0, 0.0
            for b in range(0, sc["n_cells"], 512):
                r1 = oracle.event([sc["mesh"]], [sc["convex"]], sc["face_off"], planes, refit=True, render=True, threads=1,
                                  cell_begin=b, cell_end=min(b + 64, sc["n_cells"]))
                t1_frag += int(r1["frag_ids"].shape[0]); t1_sec += r1["seconds"]
            ncpu = os.cpu_count() or 1
            rall = oracle.event([sc["mesh"]], [sc["convex"]], sc["face_off"], planes, refit=True, render=True, threads=ncpu)
            out["cpu_baseline"]["other_thread_counts"] = {
                "1": {"value": t1_frag / max(t1_sec, 1e-9), "sample": "cells [512k, 512k+64), %.2f s" % t1_sec},
                str(ncpu): {"value": nref / rall["seconds"], "sample": "whole event, %.2f s" % rall["seconds"]}}
            out["parity_check"] = {"fragments_gpu": total_frag, "fragments_cpu": nref,
                                   "mesh_nbr_equal": bool(np.array_equal(parts[0][1]["mesh_nbr"], ref["mesh_nbr"])),
                                   "idx_equal": bool(np.array_equal(parts[0][1]["idx"], ref["idx"]))}
        def _clean(o):      # (strict JSON: a figure that was not measured is null, not NaN)
            if isinstance(o, dict):
                return {k: _clean(v) for k, v in o.items()}
            if isinstance(o, (list, tuple)):
                return [_clean(v) for v in o]
            if isinstance(o, float) and (o != o or o in (float("inf"), float("-inf"))):
                return None
            return o
        print(json.dumps(_clean(out)), file=json_out, flush=True)
    for e in engs:
        e.close()
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
