#!/bin/bash
# Fuzz with the contexts told that six of them share the GPU (SURTR_EVENTS_IN_FLIGHT=6: what bench.py's contexts run with --
# lean kernels for events of any size, two polling catchers, a quarter of the faces scratch tier).  Run through gpurun.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/fuzz
export SURTR_EVENTS_IN_FLIGHT=6
(python scripts/fuzz_refracture_gpu.py 900 919191 > gpurun_out/fuzz/r4g_refr_919191.log 2>&1; tail -1 gpurun_out/fuzz/r4g_refr_919191.log) &
(python scripts/fuzz_gpu.py 2500 474747 > gpurun_out/fuzz/r4g_fuzz_474747.log 2>&1; tail -1 gpurun_out/fuzz/r4g_fuzz_474747.log) &
(python scripts/fuzz_gpu.py 2500 585858 > gpurun_out/fuzz/r4g_fuzz_585858.log 2>&1; tail -1 gpurun_out/fuzz/r4g_fuzz_585858.log) &
(SURTR_FACES_TIER_HE=512 python scripts/fuzz_gpu.py 2000 696969 > gpurun_out/fuzz/r4g_fuzz_tier512_696969.log 2>&1; tail -1 gpurun_out/fuzz/r4g_fuzz_tier512_696969.log) &
wait
