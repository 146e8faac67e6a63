#!/bin/bash
# Per-kernel LDS / VGPR / scratch usage of the device code (compiles surtr_hip.hip to an object with the build's flags).
set -e
cd "$(dirname "$0")/.."
mkdir -p build_tmp/res
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -fPIC --cuda-device-only -c "$@" \
    -o build_tmp/res/dev.o surtr_amd/csrc/surtr_hip.hip
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=build_tmp/res/dev.o \
    --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=build_tmp/res/dev.co
/opt/rocm/lib/llvm/bin/llvm-readelf --notes build_tmp/res/dev.co | python3 -c '
import sys, re
cur = {}
for line in sys.stdin:
    m = re.match(r"\s+[-]?\s*\.(name|group_segment_fixed_size|private_segment_fixed_size|vgpr_count|agpr_count|sgpr_count|vgpr_spill_count):\s+(\S+)", line)
    if not m: continue
    k, v = m.groups()
    if k == "name" and not v.startswith("_Z") and not v.startswith("k_"): continue
    cur[k] = v
    if k == "vgpr_spill_count" or len(cur) == 7:
        pass
    if set(cur) >= {"name", "group_segment_fixed_size", "private_segment_fixed_size", "vgpr_count", "vgpr_spill_count"}:
        print("%-60s lds %6s scratch %5s vgpr %4s agpr %4s spill %s" % (cur["name"][:60], cur["group_segment_fixed_size"], cur["private_segment_fixed_size"], cur["vgpr_count"], cur.get("agpr_count", "-"), cur["vgpr_spill_count"]))
        cur = {}
'
