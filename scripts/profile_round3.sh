#!/bin/bash
# Third measurement pass (run through gpurun): the bench line, rocprofv3 kernel statistics of the same command, and of the same
# command with one event at a time (--in-flight 1: the kernels alone on the GPU, where the live HIP-event timing and the
# profiler's average must agree closely); then fuzz runs on the same build.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03
mkdir -p $OUT gpurun_out/fuzz
python bench.py > $OUT/bench.json 2> $OUT/bench.err; python -c "import json; d=json.load(open('$OUT/bench.json')); print(d['value'], d['ms_per_step'], d['ms_per_fracture_event'], d['roofline'])"
rm -rf $OUT/stats3 $OUT/stats4
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats3 -- python bench.py --steps 30 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/rocprof.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats4 -- python bench.py --steps 30 --no-cpu-baseline --in-flight 1 > $OUT/bench_under_rocprof_inflight1.json 2> $OUT/rocprof1.err
find $OUT/stats3 $OUT/stats4 -name "*kernel_stats.csv"
python -c "
import json
for f in ('bench_under_rocprof.json', 'bench_under_rocprof_inflight1.json'):
    d = json.load(open('$OUT/' + f)); print(f, d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['avg_launch_ms_alone'])"
(python scripts/fuzz_refracture_gpu.py 800 27182818 > gpurun_out/fuzz/r3_final_refr_27182818.log 2>&1; tail -1 gpurun_out/fuzz/r3_final_refr_27182818.log) &
(python scripts/fuzz_refracture_gpu.py 800 13579 > gpurun_out/fuzz/r3_final_refr_13579.log 2>&1; tail -1 gpurun_out/fuzz/r3_final_refr_13579.log) &
(SURTR_WAVE=1 python scripts/fuzz_gpu.py 3000 5550123 > gpurun_out/fuzz/r3_final_fuzz_wave_5550123.log 2>&1; tail -1 gpurun_out/fuzz/r3_final_fuzz_wave_5550123.log) &
(python scripts/fuzz_gpu.py 2000 99001 > gpurun_out/fuzz/r3_final_fuzz_99001.log 2>&1; tail -1 gpurun_out/fuzz/r3_final_fuzz_99001.log) &
wait
