"""Probe: static vs work-queue (dynamic) item assignment in the stream kernel, interleaved A/B."""
import os, sys, statistics
sys.path.insert(0, ".")
os.environ["PGENHIP_WIDE_NT"] = "1"; os.environ["PGENHIP_WIDE_STREAM"] = "7"
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
                os.environ["PGENHIP_WIDE_DYN"], os.environ["PGENHIP_WIDE_BLOCKS_PER_CU"] = str(c[0]), str(c[1])
                eng.timer_start(); eng.decode_emit(recs, v, out=out, kernel=4); ms = eng.timer_stop()
                if r: ts[c].append(ms)
        for c in cfgs:
            med = statistics.median(ts[c])
            print(f"N={n} V={v} dyn/bpc={c}: med {med:.3f} min {min(ts[c]):.3f} ms  {alg/med/1e9:.3f} TB/s", flush=True)

if __name__ == "__main__":
    cfgs = [(0, 8), (0, 3), (1, 2), (1, 3), (1, 4), (1, 8), (1, 16)]
    ab(2504, 1_103_547, cfgs)
    ab(500_000, 6_000, cfgs)
    ab(2504, 4 * 1_103_547, [(0, 8), (1, 3), (1, 8)], rounds=4)
