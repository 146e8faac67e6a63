#!/bin/bash
# Long fuzz on the round's last build (run through gpurun): four processes side by side, logs under gpurun_out/fuzz/.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/fuzz
(python scripts/fuzz_refracture_gpu.py 1600 717171 > gpurun_out/fuzz/r3_end_refr_717171.log 2>&1; tail -1 gpurun_out/fuzz/r3_end_refr_717171.log) &
(python scripts/fuzz_refracture_gpu.py 1600 24680 > gpurun_out/fuzz/r3_end_refr_24680.log 2>&1; tail -1 gpurun_out/fuzz/r3_end_refr_24680.log) &
(SURTR_WAVE=1 python scripts/fuzz_gpu.py 6000 535354 > gpurun_out/fuzz/r3_end_fuzz_wave_535354.log 2>&1; tail -1 gpurun_out/fuzz/r3_end_fuzz_wave_535354.log) &
(SURTR_WAVE=1 SURTR_WAVE_BIG=1 python scripts/fuzz_gpu.py 4000 20261007 > gpurun_out/fuzz/r3_end_fuzz_wavebig_20261007.log 2>&1; tail -1 gpurun_out/fuzz/r3_end_fuzz_wavebig_20261007.log) &
wait
