#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03c
mkdir -p $OUT gpurun_out/fuzz
python -m pytest tests -x -q -m gpu > $OUT/tests_full.log 2>&1; tail -3 $OUT/tests_full.log
python -m pytest tests/test_gpu_parity.py::test_errors_are_reported_on_the_gpu -x -q -m gpu > $OUT/tests3.log 2>&1; tail -1 $OUT/tests3.log
(SURTR_WAVE=1 python scripts/fuzz_gpu.py 2500 1123581321 > gpurun_out/fuzz/r3_fused_fuzz_wave_1123581321.log 2>&1; tail -1 gpurun_out/fuzz/r3_fused_fuzz_wave_1123581321.log) &
(SURTR_WAVE=1 SURTR_WAVE_BIG=1 python scripts/fuzz_gpu.py 1500 20261005 > gpurun_out/fuzz/r3_fused_fuzz_wavebig_20261005.log 2>&1; tail -1 gpurun_out/fuzz/r3_fused_fuzz_wavebig_20261005.log) &
(SURTR_WAVE=1 SURTR_WWALK0=1 python scripts/fuzz_gpu.py 1500 777002 > gpurun_out/fuzz/r3_fused_fuzz_wave_walk1_777002.log 2>&1; tail -1 gpurun_out/fuzz/r3_fused_fuzz_wave_walk1_777002.log) &
(python scripts/fuzz_refracture_gpu.py 700 161803 > gpurun_out/fuzz/r3_fused_refr_161803.log 2>&1; tail -1 gpurun_out/fuzz/r3_fused_refr_161803.log) &
wait
