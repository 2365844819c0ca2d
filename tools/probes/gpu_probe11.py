import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, pgen_rs_amd
from pgen_rs_amd import _capi
def ab(n, v, kept, label):
    with pgen_rs_amd.GtEngine(n, kept_idx=kept, device=0) as eng:
        recs = eng.synth_records(v)
        out = torch.empty(v * eng.gt_row_bytes, dtype=torch.uint8, device="cuda:0")
        for env in ("0", "1"):
            os.environ["PGENHIP_SCAN_PICK_DENSE"] = env
            ts = []
            for r in range(6):
                eng.timer_start(); eng.decode_emit(recs, v, out=out, kernel=_capi.KERNEL_SCAN); ms = eng.timer_stop()
                if r: ts.append(ms)
            med = statistics.median(ts); alg = v * (eng.record_size + eng.gt_row_bytes)
            print(f"N={n} V={v} K={eng.kept_count} ({label}) pick_dense={env}: {med:.3f} ms  {alg/med/1e9:.3f} TB/s ({alg/med/8e9:.3f})", flush=True)
rng = np.random.default_rng(1)
n = 500_000
ab(n, 3000, np.arange(0, n - 7, dtype=np.uint32), "all but 7")
ab(n, 3000, np.sort(rng.choice(n, size=int(n * 0.9), replace=False)).astype(np.uint32), "90%")
ab(n, 3000, np.sort(rng.choice(n, size=int(n * 0.8), replace=False)).astype(np.uint32), "80%")
