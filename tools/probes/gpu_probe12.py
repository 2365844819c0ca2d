"""Probe: gather kernels vs segment pick kernel around their hand-over densities (N = 500 000)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, pgen_rs_amd
from pgen_rs_amd import _capi
def ab(n, v, kept, label):
    with pgen_rs_amd.GtEngine(n, kept_idx=kept, device=0) as eng:
        recs = eng.synth_records(v)
        out = torch.empty(v * eng.gt_row_bytes, dtype=torch.uint8, device="cuda:0")
        for name, env in (("gather3", {"PGENHIP_SCAN_SUPER": "1"}), ("pick", {"PGENHIP_SCAN_SUPER": "0"}), ("ctz", {"PGENHIP_SCAN_SUPER": "0", "PGENHIP_SCAN_PICK": "0"})):
            for k in ("PGENHIP_SCAN_PICK", "PGENHIP_SCAN_SUPER"):
                os.environ.pop(k, None)
            os.environ.update(env)
            ts = []
            for r in range(6):
                eng.timer_start(); eng.decode_emit(recs, v, out=out, kernel=_capi.KERNEL_SCAN); ms = eng.timer_stop()
                if r: ts.append(ms)
            med = statistics.median(ts); alg = v * (eng.record_size + eng.gt_row_bytes)
            print(f"N={n} V={v} K={eng.kept_count} ({label}) {name}: {med:.3f} ms  {alg/med/1e9:.3f} TB/s ({alg/med/8e9:.3f})", flush=True)
rng = np.random.default_rng(2)
n = 500_000
for frac in (0.003, 0.006, 0.01, 0.02):
    ab(n, 60000, np.sort(rng.choice(n, size=int(n * frac), replace=False)).astype(np.uint32), f"{frac*100:g}%")
