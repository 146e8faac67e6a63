#!/bin/bash
# Round-4 measurement pass on the GPU box (run through gpurun): the bench line, rocprofv3 kernel statistics of the same command
# (the default number of events in flight, and one at a time: there the live HIP-event timing and the profiler's average must agree), the PMC
# passes (-> profiles/traffic.json for this build), phase stamps of the pre-pass and of the record clipper, side benchmarks.
# Everything lands under gpurun_out/r04/; the summaries that are evidence are then copied to profiles/ by hand.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r04
mkdir -p $OUT
python bench.py > $OUT/bench.json 2> $OUT/bench.err; python -c "import json; d=json.load(open('$OUT/bench.json')); print(d['value'], d['ms_per_step'], d['ms_per_fracture_event'], d['roofline']['frac'], d['cpu_baseline']['value'])"
rm -rf $OUT/stats3 $OUT/stats1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats3 -- python bench.py --steps 30 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/rocprof.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats1 -- python bench.py --steps 30 --no-cpu-baseline --in-flight 1 > $OUT/bench_under_rocprof_inflight1.json 2> $OUT/rocprof1.err
find $OUT/stats3 $OUT/stats1 -name "*kernel_stats.csv"
bash scripts/pmc.sh > $OUT/pmc_run.log 2>&1
python scripts/pmc_summary.py gpurun_out/pmc $OUT/pmc_summary.json profiles/r02_fetch_calibration.json > $OUT/pmc_summary.txt 2>&1; grep "k_prep\|k_clip_pairs_wave\|traffic.json" $OUT/pmc_summary.txt
cp profiles/traffic.json $OUT/traffic.json
python bench.py --no-cpu-baseline > $OUT/bench_with_traffic.json 2>/dev/null; python -c "import json; d=json.load(open('$OUT/bench_with_traffic.json')); print(d['roofline']['traffic'], d['roofline_front_half'])"
if [ -f build_tmp/libsurtr_hip_stamp.so ]; then
  python scripts/stamps.py build_tmp/libsurtr_hip_stamp.so 3 > $OUT/prep_stamps.txt 2>&1
  python scripts/stamps_wave.py build_tmp/libsurtr_hip_stamp.so > $OUT/wave_stamps.txt 2>&1
  python scripts/wave_need.py build_tmp/libsurtr_hip_stamp.so > $OUT/wave_need.txt 2>&1
  python scripts/stamps_convex.py build_tmp/libsurtr_hip_stamp.so 2>&1 | grep -v amdgpu.ids > $OUT/convex_stamps.txt
fi
# kernel timelines of ONE small event (configs[1], configs[2], a 512-cell block of configs[3]): where the time between the kernels goes
for w in blob64 blob1024 block8; do
  rm -rf $OUT/tl_$w
  rocprofv3 --kernel-trace --output-format csv -d $OUT/tl_$w -- python scripts/timeline.py run $w > $OUT/tl_$w.run 2>&1
  python scripts/timeline.py report $OUT/tl_$w > $OUT/timeline_$w.txt 2>&1
  python scripts/timeline.py run $w 2>&1 | grep "^event" | tail -3 >> $OUT/timeline_$w.txt
done
python scripts/cvx_diag.py 2>&1 | grep -v amdgpu.ids > $OUT/one_wave_take_rates.txt; cat $OUT/one_wave_take_rates.txt
python scripts/bench_slices_inflight.py 8 2>&1 | grep -v amdgpu.ids | tail -2 > $OUT/slices_inflight.txt; cat $OUT/slices_inflight.txt
for n in 2 4 8; do python scripts/bench_slices.py $n > $OUT/slices${n}.log 2>&1; tail -1 $OUT/slices${n}.log; done
cat $OUT/slices2.log $OUT/slices4.log $OUT/slices8.log | grep -v amdgpu.ids > $OUT/slices.txt
python scripts/bench_cfg23.py 2>&1 | grep -v amdgpu.ids > $OUT/cfg23.log; cut -c1-90 $OUT/cfg23.log
python scripts/bench_cfg5.py 2>&1 | grep -v amdgpu.ids > $OUT/cfg5.log; tail -3 $OUT/cfg5.log | cut -c1-200
python scripts/bench_regroup.py 2>&1 | grep -v amdgpu.ids > $OUT/regroup.log; cat $OUT/regroup.log
rm -f $OUT/switches.txt
for e in "SURTR_SMALL=1" "SURTR_REC=0" "SURTR_PREP_SORTED=0" "SURTR_FRONT_PAR=1"; do env $e python scripts/bench_event.py 2>&1 | grep -v amdgpu.ids >> $OUT/switches.txt; done; python scripts/bench_event.py 2>&1 | grep -v amdgpu.ids >> $OUT/switches.txt; cut -c1-120 $OUT/switches.txt
