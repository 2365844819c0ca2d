"""Probe: kept-subset kernels (scan vs list gather) at config-5 geometry and other densities."""
import os, sys, statistics
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
import numpy as np
import torch
import pgen_rs_amd
from pgen_rs_amd import _capi
from pgen_rs_amd.synth import keep_indices

def ab(n, v, kept, label, kernels=(1, 3), rounds=7):
    with pgen_rs_amd.GtEngine(n, kept_idx=kept, device=0) as eng:
        recs = eng.synth_records(v)
        out = torch.empty(v * eng.gt_row_bytes, dtype=torch.uint8, device="cuda:0")
        ts = {k: [] for k in kernels}
        for r in range(rounds + 1):
            for k in kernels:
                eng.timer_start()
                eng.decode_emit(recs, v, out=out, kernel=k)
                ms = eng.timer_stop()
                if r: ts[k].append(ms)
        alg = v * (eng.record_size + eng.gt_row_bytes)
        rd = v * eng.record_size
        for k in kernels:
            med = statistics.median(ts[k])
            print(f"N={n} V={v} K={eng.kept_count} ({label}) kernel={'rows/gather' if k == 1 else 'scan'}: med {med:.3f} ms  alg {alg/med/1e9:.3f} TB/s  read-only {rd/med/1e9:.3f} TB/s  {v*n/med/1e6:.0f} Ggt/s decoded", flush=True)

if __name__ == "__main__":
    n = 500_000
    ab(n, 8_000, keep_indices(n, modulus=100), "1% splitmix")
    ab(n, 8_000, keep_indices(n, modulus=1000), "0.1%")
    ab(n, 6_000, keep_indices(n, modulus=10), "10%")
    ab(n, 3_000, keep_indices(n, modulus=2), "50%")
    ab(n, 2_000, np.arange(0, n - 7, dtype=np.uint32), "all but 7")
    n = 2504
    ab(n, 1_103_547, keep_indices(n, modulus=100), "1%")
    ab(n, 1_103_547, keep_indices(n, modulus=2), "50%")
