#!/bin/bash
# Fuzz on the round's last build (run through gpurun): four processes side by side, logs under gpurun_out/fuzz/.
# (small events take the side-by-side front, the twin-slot one-wave clipper and the ranked face loops by default)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/fuzz
(python scripts/fuzz_refracture_gpu.py 900 818181 > gpurun_out/fuzz/r4_f2_refr_818181.log 2>&1; tail -1 gpurun_out/fuzz/r4_f2_refr_818181.log) &
(python scripts/fuzz_gpu.py 2500 464646 > gpurun_out/fuzz/r4_f2_fuzz_464646.log 2>&1; tail -1 gpurun_out/fuzz/r4_f2_fuzz_464646.log) &
(SURTR_WAVE=1 python scripts/fuzz_gpu.py 2500 575757 > gpurun_out/fuzz/r4_f2_fuzz_wave_575757.log 2>&1; tail -1 gpurun_out/fuzz/r4_f2_fuzz_wave_575757.log) &
(SURTR_FRONT_PAR=0 SURTR_WAVE=1 SURTR_WAVE_BIG=1 python scripts/fuzz_gpu.py 2000 20261005 > gpurun_out/fuzz/r4_f2_fuzz_wavebig_20261005.log 2>&1; tail -1 gpurun_out/fuzz/r4_f2_fuzz_wavebig_20261005.log) &
wait
