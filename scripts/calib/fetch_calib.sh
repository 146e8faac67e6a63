#!/bin/bash
# FETCH_SIZE calibration (see fetch_calib.hip).  Run on the GPU box from the repo root: bash scripts/calib/fetch_calib.sh
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/calib
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o gpurun_out/calib/fetch_calib scripts/calib/fetch_calib.hip
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/calib/fetch -- ./gpurun_out/calib/fetch_calib > gpurun_out/calib/bytes.json
python3 - <<'PY'
import csv, glob, json
want = json.load(open("gpurun_out/calib/bytes.json"))
f = sorted(glob.glob("gpurun_out/calib/fetch/*/*counter_collection.csv"))[-1]
got = {}
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == "FETCH_SIZE":
        got[r["Kernel_Name"].split("(")[0]] = float(r["Counter_Value"]) * 1024.0
out = {"source": "scripts/calib/fetch_calib.sh: rocprofv3 --pmc FETCH_SIZE on single-width reads of a 3 GiB buffer, MI355X",
       "k16_dwordx4_per_lane": got["k16"] / want["k16"], "k4_dword_per_lane": got["k4"] / want["k4"],
       "k12_three_dwords_stride12": got["k12"] / want["k12"],
       "g4_random_dword_vs_useful_bytes": got["g4"] / want["g4_useful"], "g4_random_dword_vs_64B_lines": got["g4"] / want["g4_lines64"]}
json.dump(out, open("gpurun_out/calib/fetch_calibration.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
