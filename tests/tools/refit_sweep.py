#!/usr/bin/env python3
"""Wide sweep of tests/test_device_refit.py::test_refit_and_rebuild_random_sequences_vs_oracle: seeds [first, last), each a random
scene / mover set / settings / set of rebuild frames; the device refit (+ LBVH rebuilds) against the oracle fed the very trees the
device holds, every buffer of every frame bit for bit.  Prints one JSON line.   Usage: python tests/tools/refit_sweep.py 0 200"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("HIKARI_HIP_DEFAULT_CTX_FLAGS", "32")  # the suite's bit-exact traversal order (tests/conftest.py)
import test_device_refit as T

first, last = int(sys.argv[1]), int(sys.argv[2])
bad, t0 = {}, time.time()
for seed in range(first, last):
    try:
        T.test_refit_and_rebuild_random_sequences_vs_oracle.__wrapped__(seed) if hasattr(T.test_refit_and_rebuild_random_sequences_vs_oracle, "__wrapped__") else T.test_refit_and_rebuild_random_sequences_vs_oracle(seed)
    except AssertionError as e:
        bad[seed] = str(e)[:200]
print(json.dumps({"seeds": [first, last], "mismatching_seeds": len(bad), "first": dict(list(bad.items())[:4]), "seconds": round(time.time() - t0, 1)}))
