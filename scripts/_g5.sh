#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03b
mkdir -p $OUT
for w in 2 4 6 8 12 16 32; do
  SURTR_WWALK0=$w python bench.py --no-cpu-baseline --in-flight 1 --steps 40 > $OUT/bench_w$w.json 2>/dev/null
  python -c "import json; d=json.load(open('$OUT/bench_w$w.json')); print('walk0', $w, 'ms/step', round(d['ms_per_step'],4), 'clip', round(d['roofline']['avg_launch_ms'],4))"
done
