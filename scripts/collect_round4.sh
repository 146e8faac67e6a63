#!/bin/bash
# Copies the summaries of scripts/profile_round4.sh (gpurun_out/r04, merged back by gpurun) into profiles/ under their round-4 names.
cd "$(dirname "$0")/.."
R=gpurun_out/r04; P=profiles
cp $R/bench.json $P/r04_bench_first_run.json
cp $R/bench_with_traffic.json $P/r04_bench.json
cp $R/bench_under_rocprof.json $P/r04_bench_under_rocprof.json
cp $R/bench_under_rocprof_inflight1.json $P/r04_bench_under_rocprof_inflight1.json
cp $(ls -t $R/stats3/*/*kernel_stats.csv | head -1) $P/r04_kernel_stats.csv
cp $(ls -t $R/stats1/*/*kernel_stats.csv | head -1) $P/r04_kernel_stats_inflight1.csv
cp $R/pmc_summary.json $P/r04_pmc_summary.json
cp $R/traffic.json $P/traffic.json
for f in prep_stamps wave_stamps wave_need convex_stamps one_wave_take_rates slices slices_inflight; do cp $R/$f.txt $P/r04_$f.txt; done
for w in blob64 blob1024 block8; do cp $R/timeline_$w.txt $P/r04_timeline_$w.txt; done
(echo "# scripts/bench_cfg23.py (configs[1], configs[2]; kernel profiling events on)"; cat $R/cfg23.log; echo; echo "# scripts/bench_cfg5.py (configs[4]: 256 cells -> fragments -> pieces x 32 cells)"; cat $R/cfg5.log; echo; echo "# scripts/bench_regroup.py"; cat $R/regroup.log; echo; echo "# scripts/bench_event.py with switches"; cat $R/switches.txt) > $P/r04_side_benchmarks.txt
python -c "import bench, json; t = json.load(open('profiles/traffic.json')); print('traffic.json build', t['build_id'], 'sources', bench.kernel_build_id())"
