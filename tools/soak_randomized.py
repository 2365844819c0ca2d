"""Soak run of tests/test_gt_parity_gpu.py::test_randomized_differential_auto_dispatch over 400 more seeds (16 000 random calls against
the oracle, ~70 s on an MI355X).  Test infrastructure: uses the oracle like the test it drives.  Last run of the round: 0 failures."""
import sys, pytest
sys.path.insert(0, "tests"); sys.path.insert(0, "oracle"); sys.path.insert(0, ".")
import importlib
m = importlib.import_module("test_gt_parity_gpu")
bad = 0
for seed in range(1000, 1400):
    try:
        m.test_randomized_differential_auto_dispatch(seed)
    except Exception as e:
        bad += 1
        print("FAIL", seed, str(e)[:300], flush=True)
        if bad > 5:
            break
    if seed % 50 == 0:
        print("seed", seed, "ok so far", flush=True)
print("done, failures:", bad)
