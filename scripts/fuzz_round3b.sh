mkdir -p gpurun_out/fuzz
(python scripts/fuzz_refracture_gpu.py 1400 90210 > gpurun_out/fuzz/r3_refr_90210_1400.log 2>&1; tail -1 gpurun_out/fuzz/r3_refr_90210_1400.log) &
(python scripts/fuzz_refracture_gpu.py 700 13579 > gpurun_out/fuzz/r3_refr_13579_700.log 2>&1; tail -1 gpurun_out/fuzz/r3_refr_13579_700.log) &
(python scripts/fuzz_refracture_gpu.py 120 555002 > gpurun_out/fuzz/r3_refr_555002_120.log 2>&1; tail -1 gpurun_out/fuzz/r3_refr_555002_120.log) &
(python scripts/fuzz_gpu.py 1500 777001 > gpurun_out/fuzz/r3_fuzz_777001_1500.log 2>&1; tail -1 gpurun_out/fuzz/r3_fuzz_777001_1500.log) &
wait
