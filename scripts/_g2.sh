python scripts/bench_big.py 500 200 4096 0 2>&1 | tail -1 | cut -c1-420
python scripts/bench_big.py 700 300 4096 0 2>&1 | tail -1 | cut -c1-420
SURTR_WAVE_BIG=0 python scripts/bench_big.py 700 300 4096 0 2>&1 | tail -1 | cut -c1-420
python scripts/bench_big.py 1000 500 4096 8 2>&1 | tail -2 | cut -c1-420
