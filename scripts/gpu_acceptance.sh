#!/bin/bash
# What the driver runs at round end, in one go (run through gpurun): the GPU test tier, smoke(), the default bench line (timed),
# and a one-rank rehearsal of the N > 1 code path of bench.py (process group, all-gather, both scaling modes).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/accept
mkdir -p $OUT
python -m pytest tests -x -q -m gpu > $OUT/tests_gpu.log 2>&1; tail -2 $OUT/tests_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
T0=$(date +%s); python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench.py default: $(( $(date +%s) - T0 )) s wall"; cut -c1-300 $OUT/bench.json
python bench.py --force-dist --no-cpu-baseline --steps 10 > $OUT/bench_force_dist.json 2> $OUT/bench_force_dist.err; tail -c 600 $OUT/bench_force_dist.json; echo
