"""Probe: stream-span kernel (5) vs row-item work-queue kernel (4), interleaved A/B at HBM-resident sizes."""
import os, sys, statistics
sys.path.insert(0, ".")
import torch
import pgen_rs_amd

def ab(n, v, cfgs, rounds=10):
    with pgen_rs_amd.GtEngine(n, device=0) as eng:
        recs = eng.synth_records(v)
        out = torch.empty(v * eng.gt_row_bytes, dtype=torch.uint8, device="cuda:0")
        alg = v * (eng.record_size + eng.gt_row_bytes)
        ts = {c: [] for c in cfgs}
        for r in range(rounds + 1):
            for c in cfgs:
                os.environ["PGENHIP_WIDE_BLOCKS_PER_CU"] = str(c[1]); os.environ["PGENHIP_WIDE_NT"] = str(c[2])
                eng.timer_start(); eng.decode_emit(recs, v, out=out, kernel=c[0]); ms = eng.timer_stop()
                if r: ts[c].append(ms)
        for c in cfgs:
            med = statistics.median(ts[c])
            print(f"N={n} V={v} kernel/bpc/nt={c}: med {med:.3f} min {min(ts[c]):.3f} ms  {alg/med/1e9:.3f} TB/s", flush=True)

if __name__ == "__main__":
    cfgs = [(4, 3, 1), (5, 2, 1), (5, 3, 1), (5, 4, 1), (5, 2, 0), (5, 3, 0)]
    ab(2504, 1_103_547, cfgs)
    ab(500_000, 6_000, cfgs)
    ab(2504, 4 * 1_103_547, [(4, 3, 1), (5, 2, 1), (5, 3, 1)], rounds=4)
