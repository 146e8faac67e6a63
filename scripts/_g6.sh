#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03c
mkdir -p $OUT
python -m pytest tests/test_record_clipper.py tests/test_gpu_parity.py -x -q -m gpu > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log
python scripts/stamps_wave.py build_tmp/libsurtr_hip_stamp.so > $OUT/wave_stamps.txt 2>&1; tail -26 $OUT/wave_stamps.txt
python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; python -c "import json; d=json.load(open('$OUT/bench.json')); print(d['ms_per_step'], d.get('ms_per_fracture_event'), d['roofline']['avg_launch_ms'])"
