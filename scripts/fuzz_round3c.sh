mkdir -p gpurun_out/fuzz
(python scripts/fuzz_refracture_gpu.py 900 31337 > gpurun_out/fuzz/r3_final_refr_31337.log 2>&1; tail -1 gpurun_out/fuzz/r3_final_refr_31337.log) &
(python scripts/fuzz_refracture_gpu.py 900 4242 > gpurun_out/fuzz/r3_final_refr_4242.log 2>&1; tail -1 gpurun_out/fuzz/r3_final_refr_4242.log) &
(SURTR_WAVE=1 python scripts/fuzz_gpu.py 2500 864200 > gpurun_out/fuzz/r3_final_fuzz_wave_864200.log 2>&1; tail -1 gpurun_out/fuzz/r3_final_fuzz_wave_864200.log) &
(SURTR_WAVE=1 SURTR_WAVE_BIG=1 python scripts/fuzz_gpu.py 1500 20261004 > gpurun_out/fuzz/r3_final_fuzz_wavebig_20261004.log 2>&1; tail -1 gpurun_out/fuzz/r3_final_fuzz_wavebig_20261004.log) &
wait
