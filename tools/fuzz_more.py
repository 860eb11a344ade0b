#!/usr/bin/env python3
"""More seeds of tests/test_gpu_fuzz.py than the suite runs (ad hoc, on a GPU box):  tools/fuzz_more.py [first] [last]"""
import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import test_gpu_fuzz as F
a, b = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (48, 200)
bad = 0
for seed in range(a, b):
    try:
        F.test_fuzz_against_oracle(seed)
    except Exception:
        bad += 1
        print("seed", seed, "FAILED"); traceback.print_exc(limit=3)
        if bad > 3: break
print("fuzz seeds %d..%d: %d failed" % (a, b - 1, bad))
