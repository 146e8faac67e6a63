cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03; mkdir -p $OUT
python bench.py > $OUT/bench.json 2> $OUT/bench.err; python -c "import json; d=json.load(open('$OUT/bench.json')); print(d['ms_per_step'], d['ms_per_fracture_event'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
rm -rf $OUT/stats2; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats2 -- python bench.py --steps 30 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/rocprof.err
python -c "
import csv,glob,json
f=sorted(glob.glob('$OUT/stats2/*/*_kernel_stats.csv'))[-1]
for r in list(csv.DictReader(open(f)))[:3]: print(r['Name'].split('(')[0], r['Calls'], float(r['AverageNs'])/1e6)
d=json.load(open('$OUT/bench_under_rocprof.json')); print('under rocprof: ms/step', d['ms_per_step'], 'live avg', d['roofline']['avg_launch_ms'])
"
python scripts/stamps_wave.py build_tmp/libsurtr_hip_stamp.so > $OUT/wave_stamps.txt 2>&1; tail -22 $OUT/wave_stamps.txt | head -20
