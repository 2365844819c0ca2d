"""Ad-hoc GPU probe (not a test): interleaved timing of kernel variants on two shapes."""
import os, sys, statistics
sys.path.insert(0, ".")
import torch
import pgen_rs_amd
from pgen_rs_amd import _capi

def bench_shape(n, v, configs, rounds=7, kept=None):
    with pgen_rs_amd.GtEngine(n, kept_idx=kept, device=0) as eng:
        recs = eng.synth_records(v)
        out = torch.empty(v * eng.gt_row_bytes, dtype=torch.uint8, device="cuda:0")
        times = {c: [] for c in configs}
        for r in range(rounds + 1):
            for c in configs:
                kernel, env = c
                for k_, v_ in env: os.environ[k_] = v_
                eng.timer_start()
                eng.decode_emit(recs, v, out=out, kernel=kernel)
                ms = eng.timer_stop()
                if r: times[c].append(ms)
        bytes_alg = v * (eng.record_size + eng.gt_row_bytes)
        for c in configs:
            med = statistics.median(times[c]); mn = min(times[c])
            print(f"N={n} V={v} K={eng.kept_count} cfg={c}: med {med:.3f} ms min {mn:.3f}  {bytes_alg/med/1e9:.3f} TB/s alg  {v*n/med/1e6:.1f} Ggt/s", flush=True)

if __name__ == "__main__":
    print(torch.cuda.get_device_name(0), torch.version.hip, flush=True)
    cfgs = []
    for var in (5, 10, 1):
        for bpc in (64, 128, 256, 512, 4096):
            cfgs.append((2, (("PGENHIP_FLAT_VARIANT", str(var)), ("PGENHIP_FLAT_BLOCKS_PER_CU", str(bpc)))))
    bench_shape(2504, 1_103_547, cfgs)
    bench_shape(500_000, 2_000, cfgs)
