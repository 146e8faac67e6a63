#!/bin/bash
# Diagnostic build of the device library with per-phase cycle stamps (-DSURTR_STAMP): build_tmp/libsurtr_hip_stamp.so
set -e
cd "$(dirname "$0")/.."
mkdir -p build_tmp
S=surtr_amd/csrc
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -fPIC -shared -DSURTR_STAMP "$@" -o build_tmp/libsurtr_hip_stamp.so \
    $S/surtr_hip.hip $S/pieces_dev.hip $S/cells_dev.hip $S/mesh_dev.hip $S/regroup_dev.hip $S/host_geom.cpp $S/host_regroup.cpp
