#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03c
mkdir -p $OUT
python -m pytest tests/test_record_clipper.py -x -q -m gpu > $OUT/tests_rc.log 2>&1; tail -2 $OUT/tests_rc.log
for r in 1 2 3; do
  for L in C D; do
    SURTR_LIB=build_tmp/lib$L.so python scripts/bench_event.py 2>/dev/null | cut -c1-380
  done
done
