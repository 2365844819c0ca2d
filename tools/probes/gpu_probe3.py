"""Probe: kernels at HBM-resident sizes (input > Infinity Cache)."""
import os, sys, statistics
sys.path.insert(0, ".")
import torch
import pgen_rs_amd

def run(n, v, kernel, env, rounds=5):
    for k_, v_ in env.items(): os.environ[k_] = str(v_)
    with pgen_rs_amd.GtEngine(n, device=0) as eng:
        recs = eng.synth_records(v)
        out = torch.empty(v * eng.gt_row_bytes, dtype=torch.uint8, device="cuda:0")
        ts = []
        for r in range(rounds + 1):
            eng.timer_start()
            eng.decode_emit(recs, v, out=out, kernel=kernel)
            ms = eng.timer_stop()
            if r: ts.append(ms)
        med = statistics.median(ts)
        alg = v * (eng.record_size + eng.gt_row_bytes)
        print(f"N={n} V={v} kernel={kernel} {env}: med {med:.3f} ms  {alg/med/1e9:.3f} TB/s alg", flush=True)
    del recs, out
    torch.cuda.empty_cache()

if __name__ == "__main__":
    for n, v in ((2504, 1_103_547), (500_000, 6_000)):
        run(n, v, 2, {"PGENHIP_FLAT_VARIANT": 1, "PGENHIP_FLAT_BLOCKS_PER_CU": 64})
        for ns in (0, 3, 7):
            for bpc in (2, 3, 4, 6, 8):
                for nt in (0, 1):
                    run(n, v, 4, {"PGENHIP_WIDE_STREAM": ns, "PGENHIP_WIDE_BLOCKS_PER_CU": bpc, "PGENHIP_WIDE_NT": nt})
