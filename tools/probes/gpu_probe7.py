"""Diagnostic: per-wave start/end stamps of the symmetric wide kernel (PGENHIP_DEBUG_TIMES=1)."""
import os, sys
sys.path.insert(0, ".")
os.environ["PGENHIP_DEBUG_TIMES"] = "1"
os.environ["PGENHIP_WIDE_STREAM"] = "0"
os.environ["PGENHIP_WIDE_NT"] = "1"
import numpy as np, torch
import pgen_rs_amd
n, v = 2504, 1_103_547
for bpc in (6, 8, 64, 1024):
    os.environ["PGENHIP_WIDE_BLOCKS_PER_CU"] = str(bpc)
    with pgen_rs_amd.GtEngine(n, device=0) as eng:
        recs = eng.synth_records(v)
        out = torch.empty(v * eng.gt_row_bytes, dtype=torch.uint8, device="cuda:0")
        for rep in range(3):
            eng.timer_start(); eng.decode_emit(recs, v, out=out, kernel=4); ms = eng.timer_stop()
        d = np.fromfile("/tmp/pgenhip_times.bin", dtype=np.uint64).reshape(-1, 3)
        d = d[d[:, 0] > 0]
        t0 = d[:, 0].min()
        st = (d[:, 0] - t0) / 100.0   # us (100 MHz)
        en = (d[:, 1] - t0) / 100.0
        print(f"bpc={bpc} launch {ms:.3f} ms  waves {len(d)}  items/wave min {d[:,2].min()} max {d[:,2].max()}")
        print("  start us pctl 0/50/90/99/100:", np.percentile(st, [0, 50, 90, 99, 100]).round(1))
        print("  end   us pctl 0/1/10/50/90/100:", np.percentile(en, [0, 1, 10, 50, 90, 100]).round(1))
        dur = en - st
        print("  per-wave busy us pctl 0/50/100:", np.percentile(dur, [0, 50, 100]).round(1), " us/item median", np.median(dur / np.maximum(d[:, 2], 1)).round(2))
        # active waves over time
        edges = np.linspace(0, en.max(), 13)
        act = [int(((st <= x) & (en > x)).sum()) for x in edges]
        print("  active waves at", edges.round(0).tolist(), act)
