#!/usr/bin/env python3
"""Does the place of the OUTPUT buffer matter inside one process?  chr22 block (N = 2 504) through the stream kernel into
(a) one big allocation at several byte offsets, (b) several separate allocations.  Prints ms per launch (median of 5 x 3)."""
import statistics
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import pgen_rs_amd  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2504
v = int(sys.argv[2]) if len(sys.argv) > 2 else 1_103_547


def timeit(eng, recs, out):
    eng.decode_emit(recs, v, out=out)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(3):
            eng.decode_emit(recs, v, out=out)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / 3)
    return statistics.median(ts)


with pgen_rs_amd.GtEngine(n, device=0) as eng:
    recs = eng.synth_records(v)
    size = v * eng.gt_row_bytes
    alg = v * (eng.record_size + eng.gt_row_bytes)
    big = torch.empty(size + (256 << 20), dtype=torch.uint8, device="cuda:0")
    print(f"N={n} V={v}: {alg / 1e9:.2f} GB per launch; big allocation at 0x{big.data_ptr():x}")
    for off in (0, 128, 4096, 65536, 1 << 20, (2 << 20) + 4096, 32 << 20, 128 << 20, 0):
        ms = timeit(eng, recs, big[off : off + size])
        print(f"  offset {off:>10d}: {ms:.4f} ms  frac {alg / (ms * 1e-3) / 8e12:.3f}")
    del big
    torch.cuda.empty_cache()
    keep = []
    for i in range(5):
        out = torch.empty(size, dtype=torch.uint8, device="cuda:0")
        ms = timeit(eng, recs, out)
        print(f"  allocation {i} at 0x{out.data_ptr():x}: {ms:.4f} ms  frac {alg / (ms * 1e-3) / 8e12:.3f}")
        keep.append(out)   # hold it so the next allocation lands elsewhere
    for i, out in enumerate(keep):
        ms = timeit(eng, recs, out)
        print(f"  allocation {i} again: {ms:.4f} ms  frac {alg / (ms * 1e-3) / 8e12:.3f}")
    # the same allocations with other numbers of work-queue ranges (the launch's write fronts)
    from pgen_rs_amd import _capi
    for ranges in (1, 2, 4, 8):
        eng.tune(_capi.KNOB_WIDE_RANGES, ranges)
        print(f"  ranges = {ranges}: " + "  ".join(f"{alg / (timeit(eng, recs, out) * 1e-3) / 8e12:.3f}" for out in keep))
