#!/bin/bash
# Occupancy experiment for the record clipper (TIMING ONLY: the experimental builds DROP the pairs the clipper hands on, their
# results are invalid).  Applies scripts/exp_occupancy.patch to a copy of the kernel sources (no in-place fall-back to the
# general clipper, so that the kernel's LDS is the record clipper's alone: 119 registers, no scratch) and builds two variants:
#   build_tmp/libexp_occ3.so   36 KB record area, 50 KB of LDS: three workgroups per CU fit
#   build_tmp/libexp_occ4.so   24 KB record area, 35 KB of LDS: four fit
# Then, on the GPU box:  SURTR_WG_PER_CU=2|3|4 SURTR_LIB=build_tmp/libexp_occN.so python scripts/bench_event.py
# (the same build at two and at three workgroups per CU clips the same pairs: the ratio is what the occupancy is worth).
set -e
cd "$(dirname "$0")/.."
rm -rf build_tmp/exp_src && mkdir -p build_tmp/exp_src/surtr_amd && cp -r surtr_amd/csrc build_tmp/exp_src/surtr_amd/ && cp -r include build_tmp/exp_src/
(cd build_tmp/exp_src && patch -p1 -s < ../../scripts/exp_occupancy.patch)
S=build_tmp/exp_src/surtr_amd/csrc
SRC="$S/surtr_hip.hip $S/pieces_dev.hip $S/cells_dev.hip $S/mesh_dev.hip $S/regroup_dev.hip $S/host_geom.cpp $S/host_regroup.cpp"
COMMON="-O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -fPIC -shared"
/opt/rocm/bin/hipcc $COMMON -DSURTR_WR=2304u -DSURTR_WNL=1536u -DSURTR_EXP_OCC=3 -o build_tmp/libexp_occ3.so $SRC 2> build_tmp/exp_occ3.err
/opt/rocm/bin/hipcc $COMMON -DSURTR_WR=1536u -DSURTR_WNL=1024u -DSURTR_EXP_OCC=4 -o build_tmp/libexp_occ4.so $SRC 2> build_tmp/exp_occ4.err
ls -la build_tmp/libexp_occ*.so
