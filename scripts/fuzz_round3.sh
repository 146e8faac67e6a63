mkdir -p gpurun_out/fuzz
python -m pytest tests -m gpu -q 2>&1 | tail -2
(python scripts/fuzz_refracture_gpu.py 500 555 > gpurun_out/fuzz/r3_refr_555.log 2>&1; tail -1 gpurun_out/fuzz/r3_refr_555.log) &
(python scripts/fuzz_refracture_gpu.py 500 13579 > gpurun_out/fuzz/r3_refr_13579.log 2>&1; tail -1 gpurun_out/fuzz/r3_refr_13579.log) &
(python scripts/fuzz_refracture_gpu.py 500 24680 > gpurun_out/fuzz/r3_refr_24680.log 2>&1; tail -1 gpurun_out/fuzz/r3_refr_24680.log) &
(python scripts/fuzz_refracture_gpu.py 500 90210 > gpurun_out/fuzz/r3_refr_90210.log 2>&1; tail -1 gpurun_out/fuzz/r3_refr_90210.log) &
wait
