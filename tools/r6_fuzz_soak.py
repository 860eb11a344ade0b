#!/usr/bin/env python3
"""Round 6 soak: more seeds of the differential fuzz tests than the suite runs (all three fuzzers), after the round's changes to the
generic table quantiser, the candidate buffers and the sharded inverted index.  tools/r6_fuzz_soak.py [first] [last]"""
import os, sys, traceback, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import test_gpu_fuzz as F
a, b = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (100, 160)
out = {}
t0 = time.time()
for name in [n for n in dir(F) if n.startswith("test_fuzz")]:
    fn = getattr(F, name)
    bad = []
    for seed in range(a, b):
        try:
            fn(seed)
        except Exception:
            bad.append(seed)
            traceback.print_exc(limit=4)
            if len(bad) > 3:
                break
    out[name] = {"seeds": [a, b - 1], "failed": bad}
out["seconds"] = round(time.time() - t0, 1)
print(json.dumps(out))
