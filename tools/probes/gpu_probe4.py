"""Probe: interleaved A/B (rule 24) of the wide/stream configurations at HBM-resident sizes."""
import os, sys, statistics
sys.path.insert(0, ".")
import torch
import pgen_rs_amd

def ab(n, v, configs, rounds=12):
    with pgen_rs_amd.GtEngine(n, device=0) as eng:
        recs = eng.synth_records(v)
        out = torch.empty(v * eng.gt_row_bytes, dtype=torch.uint8, device="cuda:0")
        ts = {c: [] for c in configs}
        for r in range(rounds + 1):
            for c in configs:
                os.environ["PGENHIP_WIDE_STREAM"], os.environ["PGENHIP_WIDE_BLOCKS_PER_CU"], os.environ["PGENHIP_WIDE_NT"] = map(str, c)
                eng.timer_start()
                eng.decode_emit(recs, v, out=out, kernel=4)
                ms = eng.timer_stop()
                if r: ts[c].append(ms)
        alg = v * (eng.record_size + eng.gt_row_bytes)
        for c in configs:
            med, mn = statistics.median(ts[c]), min(ts[c])
            print(f"N={n} V={v} stream/bpc/nt={c}: med {med:.3f} min {mn:.3f} ms  {alg/med/1e9:.3f} TB/s (best {alg/mn/1e9:.3f})", flush=True)

if __name__ == "__main__":
    cfgs = [(0, 4, 1), (0, 4, 0), (0, 6, 0), (0, 8, 1), (0, 8, 0), (7, 2, 1), (7, 3, 1), (7, 3, 0), (7, 6, 1), (7, 8, 1), (3, 8, 1), (3, 4, 1)]
    ab(2504, 1_103_547, cfgs)
    ab(500_000, 6_000, cfgs)
