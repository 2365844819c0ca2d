"""Probe: list gather (general kernel, touches only the kept samples' lines) vs segment pick on very sparse keeps."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, pgen_rs_amd
from pgen_rs_amd import _capi
def ab(n, v, kept, label):
    with pgen_rs_amd.GtEngine(n, kept_idx=kept, device=0) as eng:
        recs = eng.synth_records(v)
        out = torch.empty(v * eng.gt_row_bytes, dtype=torch.uint8, device="cuda:0")
        for name, k in (("list gather", _capi.KERNEL_ROWS), ("scan family", _capi.KERNEL_SCAN)):
            ts = []
            for r in range(6):
                eng.timer_start(); eng.decode_emit(recs, v, out=out, kernel=k); ms = eng.timer_stop()
                if r: ts.append(ms)
            med = statistics.median(ts); alg = v * (eng.record_size + eng.gt_row_bytes)
            print(f"N={n} V={v} K={eng.kept_count} ({label}) {name}: {med:.3f} ms  {alg/med/1e9:.3f} TB/s alg", flush=True)
rng = np.random.default_rng(3)
n = 500_000
for frac in (0.001, 0.002, 0.004, 0.006, 0.008):
    ab(n, 60000, np.sort(rng.choice(n, size=int(n * frac), replace=False)).astype(np.uint32), f"{frac*100:g}%")
