"""GPU-vs-oracle fuzz of multi-piece events (BASELINE configs[4] shape): random first-level fracture, every fragment
re-split by its own cells in ONE event of many small pieces -- the events that take k_clip_pairs_half.
Usage: python scripts/fuzz_refracture_gpu.py [n_cases] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import oracle as O
from surtr_amd import engine as E
from helpers import assert_event_equal
from test_refracture import _refracture


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 99)
    bad = 0
    t0 = time.time()
    for case in range(n):
        n_first = int(rng.choice([6, 16, 40, 96, 200])); n_second = int(rng.choice([3, 8, 17, 32]))
        nu, nv = int(rng.integers(30, 260)), int(rng.integers(20, 200))
        try:
            c, got, ref, npieces = _refracture(E, O, n_first, n_second, nu, nv)
            assert c.status == 0
            assert_event_equal(got, ref)
            res = "ok"
        except AssertionError as e:
            res = "MISMATCH %s" % (e,); bad += 1
        print("case %d torus %dx%d first %d second %d pieces %d pairs %d frags %d %s  (%.0fs)" % (
            case, nu, nv, n_first, n_second, npieces, c.n_pairs, c.n_frag, res, time.time() - t0), flush=True)
    print("FUZZ DONE: %d cases, %d mismatches" % (n, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
