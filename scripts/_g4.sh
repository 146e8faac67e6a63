#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03b
mkdir -p $OUT gpurun_out/fuzz
(python scripts/fuzz_refracture_gpu.py 900 4242 > gpurun_out/fuzz/r3_final_refr_4242.log 2>&1; tail -1 gpurun_out/fuzz/r3_final_refr_4242.log) &
FZ=$!
python -m pytest tests/test_record_clipper.py tests/test_gpu_parity.py -x -q -m gpu > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log
python scripts/stamps_wave.py build_tmp/libsurtr_hip_stamp.so > $OUT/wave_stamps.txt 2>&1; tail -24 $OUT/wave_stamps.txt
python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; python -c "import json; d=json.load(open('$OUT/bench.json')); print(d['ms_per_step'], d.get('ms_per_fracture_event'), d['roofline'])"
python scripts/bench_chunks.py 1 2 3 4 > $OUT/chunks.log 2>&1; cat $OUT/chunks.log
wait $FZ
