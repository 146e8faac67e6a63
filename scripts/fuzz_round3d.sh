#!/bin/bash
# Long fuzz on the round's last build (run through gpurun): four processes side by side, logs under gpurun_out/fuzz/.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/fuzz
(python scripts/fuzz_refracture_gpu.py 1600 606060 > gpurun_out/fuzz/r3_last_refr_606060.log 2>&1; tail -1 gpurun_out/fuzz/r3_last_refr_606060.log) &
(python scripts/fuzz_refracture_gpu.py 1600 90210 > gpurun_out/fuzz/r3_last_refr_90210.log 2>&1; tail -1 gpurun_out/fuzz/r3_last_refr_90210.log) &
(SURTR_WAVE=1 python scripts/fuzz_gpu.py 6000 424243 > gpurun_out/fuzz/r3_last_fuzz_wave_424243.log 2>&1; tail -1 gpurun_out/fuzz/r3_last_fuzz_wave_424243.log) &
(SURTR_WAVE=1 SURTR_WAVE_BIG=1 python scripts/fuzz_gpu.py 4000 20261006 > gpurun_out/fuzz/r3_last_fuzz_wavebig_20261006.log 2>&1; tail -1 gpurun_out/fuzz/r3_last_fuzz_wavebig_20261006.log) &
wait
