#!/bin/bash
# Second measurement pass (run through gpurun): PMC counters (separate rocprofv3 --pmc passes, nothing else traced), phase stamps
# of the record clipper from the -DSURTR_STAMP build, the bench line again with the kernel timing under the timed loop's conditions.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03
mkdir -p $OUT
bash scripts/pmc.sh > $OUT/pmc.log 2>&1; tail -3 $OUT/pmc.log
python scripts/pmc_summary.py gpurun_out/pmc $OUT/pmc_summary.json profiles/r02_fetch_calibration.json > $OUT/pmc_summary.log 2>&1; tail -12 $OUT/pmc_summary.log
cp profiles/traffic.json $OUT/traffic.json
python scripts/stamps_wave.py build_tmp/libsurtr_hip_stamp.so > $OUT/wave_stamps.txt 2>&1; tail -24 $OUT/wave_stamps.txt
python bench.py > $OUT/bench.json 2> $OUT/bench.err; python -c "import json; d=json.load(open('$OUT/bench.json')); print(d['ms_per_step'], d['roofline'])"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats2 -- python bench.py --steps 30 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/rocprof.err
python scripts/bench_big.py 500 200 4096 16 > $OUT/big100k.log 2>&1; tail -2 $OUT/big100k.log | cut -c1-200
