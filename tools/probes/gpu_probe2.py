"""Probe: does a read-only prefetch pass (MALL staging) before each sub-batch beat the fused stream?"""
import os, sys, statistics
sys.path.insert(0, ".")
import torch
import pgen_rs_amd

def run(n, v, batch_mib, variant, bpc, prefetch, rounds=5):
    os.environ["PGENHIP_FLAT_VARIANT"] = str(variant)
    os.environ["PGENHIP_FLAT_BLOCKS_PER_CU"] = str(bpc)
    with pgen_rs_amd.GtEngine(n, device=0) as eng:
        recs = eng.synth_records(v)
        out = torch.empty(v * eng.gt_row_bytes, dtype=torch.uint8, device="cuda:0")
        R, S = eng.record_size, eng.gt_row_bytes
        vb = max(1, (batch_mib << 20) // R)
        recs64 = recs[: (v * R) // 8 * 8].view(torch.int64)
        ts = []
        for r in range(rounds + 1):
            eng.timer_start()
            for v0 in range(0, v, vb):
                nv = min(vb, v - v0)
                if prefetch:
                    a, b = (v0 * R) // 8, ((v0 + nv) * R) // 8
                    _ = recs64[a:b].sum()
                eng.decode_emit(recs, nv, out=out, kernel=2, records_offset=v0 * R, out_offset=v0 * S)
            ms = eng.timer_stop()
            if r: ts.append(ms)
        med = statistics.median(ts)
        alg = v * (R + S)
        print(f"N={n} V={v} batch={batch_mib}MiB var={variant} bpc={bpc} prefetch={prefetch}: med {med:.3f} ms  {alg/med/1e9:.3f} TB/s alg", flush=True)

if __name__ == "__main__":
    for n, v in ((2504, 1_103_547), (500_000, 8_000)):
        for variant, bpc in ((1, 64), (5, 64), (10, 32)):
            run(n, v, 1 << 20, variant, bpc, False)
            for batch in (32, 64, 128, 192):
                run(n, v, batch, variant, bpc, False)
                run(n, v, batch, variant, bpc, True)
