"""Probe: fixed per-launch cost vs streaming rate: t(V) = a + b*V from V = 1x and 4x the chr22 block."""
import os, sys, statistics
sys.path.insert(0, ".")
import torch
import pgen_rs_amd

def t_of(eng, recs, out, v, rounds=6):
    ts = []
    for r in range(rounds + 1):
        eng.timer_start(); eng.decode_emit(recs, v, out=out, kernel=4); ms = eng.timer_stop()
        if r: ts.append(ms)
    return statistics.median(ts)

if __name__ == "__main__":
    n, v1 = 2504, 1_103_547
    with pgen_rs_amd.GtEngine(n, device=0) as eng:
        recs = eng.synth_records(4 * v1)
        out = torch.empty(4 * v1 * eng.gt_row_bytes, dtype=torch.uint8, device="cuda:0")
        alg1 = v1 * (eng.record_size + eng.gt_row_bytes)
        for cfg in [(7, 8, 1), (7, 16, 1), (7, 32, 1), (7, 64, 1), (0, 16, 1), (0, 32, 1), (0, 64, 1), (0, 128, 1), (0, 1024, 1), (3, 32, 1)]:
            os.environ["PGENHIP_WIDE_STREAM"], os.environ["PGENHIP_WIDE_BLOCKS_PER_CU"], os.environ["PGENHIP_WIDE_NT"] = map(str, cfg)
            t1, t4 = t_of(eng, recs, out, v1), t_of(eng, recs, out, 4 * v1)
            a = (4 * t1 - t4) / 3
            b = (t4 - t1) / 3
            print(f"cfg {cfg}: t1 {t1:.3f} ms t4 {t4:.3f} ms  fixed a = {a*1e3:.0f} us  streaming rate {alg1/b/1e9:.3f} TB/s  rate@1x {alg1/t1/1e9:.3f}", flush=True)
