python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python scripts/repro_refracture_case.py 2>/dev/null | tail -1
for e in 1 2 3; do python bench.py --no-cpu-baseline --in-flight $e 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('in flight', d['config']['events_in_flight'], d['ms_per_step'], d['value'])"; done
python scripts/bench_slices.py 8 --balanced 2>&1 | tail -2 | cut -c1-150
python scripts/bench_slices.py 8 2>&1 | tail -1
