#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03c
mkdir -p $OUT
for r in 1 2 3; do
  for L in A B; do
    SURTR_LIB=build_tmp/lib$L.so python scripts/bench_event.py 2>/dev/null | cut -c1-400
  done
done
python -m pytest tests/test_record_clipper.py::test_record_clipper_big_bands_kernel_gpu tests/test_gpu_parity.py::test_errors_are_reported_on_the_gpu -x -q -m gpu > $OUT/tests2.log 2>&1; tail -3 $OUT/tests2.log
python -m pytest tests/test_gpu_parity.py::test_errors_are_reported_on_the_gpu -x -q -m gpu > $OUT/tests3.log 2>&1; tail -3 $OUT/tests3.log
