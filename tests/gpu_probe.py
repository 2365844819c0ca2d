"""Ad-hoc GPU probe (not a test): quick timing of a kernel variant on two shapes."""
import sys, time
sys.path.insert(0, ".")
import torch
import pgen_rs_amd
from pgen_rs_amd import _capi

def timeit(n, v, kernel, kept=None, reps=5):
    with pgen_rs_amd.GtEngine(n, kept_idx=kept, device=0) as eng:
        recs = eng.synth_records(v)
        out = torch.empty(v * eng.gt_row_bytes, dtype=torch.uint8, device="cuda:0")
        eng.decode_emit(recs, v, out=out, kernel=kernel); eng.wait()
        best = 1e9
        for _ in range(reps):
            eng.timer_start()
            eng.decode_emit(recs, v, out=out, kernel=kernel)
            ms = eng.timer_stop()
            best = min(best, ms)
        bytes_alg = v * (eng.record_size + eng.gt_row_bytes)
        print(f"N={n} V={v} K={eng.kept_count} kernel={kernel}: {best:.3f} ms  {bytes_alg/best/1e9*1e3/1e3:.3f} TB/s alg  {v*n/best/1e6:.1f} Ggt/s", flush=True)

if __name__ == "__main__":
    print(torch.cuda.get_device_name(0), torch.version.hip, flush=True)
    kernels = [int(k) for k in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1]
    for k in kernels:
        timeit(2504, 200_000, k)
        timeit(500_000, 2_000, k)
