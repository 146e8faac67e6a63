#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu 2>&1 | tail -1
for r in 1 2; do for L in H K; do SURTR_LIB=build_tmp/lib$L.so python scripts/bench_event.py 2>/dev/null | cut -c1-70,140-420; done; done
python scripts/bench_big.py 500 200 4096 16 2>&1 | tail -2 | cut -c1-420
python scripts/bench_big.py 700 300 4096 12 2>&1 | tail -2 | cut -c1-420
python scripts/bench_big.py 1000 500 4096 8 2>&1 | tail -2 | cut -c1-420
