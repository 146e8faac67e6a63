#!/bin/bash
# One measurement pass on the GPU box (run through gpurun): bench line, rocprofv3 kernel statistics of the same command, the PMC
# passes, side benchmarks.  Everything lands under gpurun_out/r03/; the summaries that are evidence are then copied to profiles/.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03
mkdir -p $OUT
python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 400 $OUT/bench.json; echo
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python bench.py --steps 30 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/rocprof.err
find $OUT/stats -name "*kernel_stats.csv" | head -2
python bench.py --in-flight 1 --no-cpu-baseline > $OUT/bench_inflight1.json 2>/dev/null
for n in 2 4 8; do
  python scripts/bench_slices.py $n > $OUT/slices${n}_equal.log 2>&1; tail -1 $OUT/slices${n}_equal.log
  python scripts/bench_slices.py $n --balanced > $OUT/slices${n}_balanced.log 2>&1; tail -1 $OUT/slices${n}_balanced.log
done
cat $OUT/slices2_equal.log $OUT/slices2_balanced.log $OUT/slices4_equal.log $OUT/slices4_balanced.log $OUT/slices8_equal.log $OUT/slices8_balanced.log > $OUT/slices.txt
python scripts/bench_cfg23.py > $OUT/cfg23.log 2>&1; cut -c1-80 $OUT/cfg23.log
python scripts/bench_cfg5.py > $OUT/cfg5.log 2>&1; tail -3 $OUT/cfg5.log | cut -c1-200
