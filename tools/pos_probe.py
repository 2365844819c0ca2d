#!/usr/bin/env python3
"""Does WHERE a chunk of records lies inside one big allocation change how fast the sparse-keep path reads it?
Times pgenhip_decode_emit on chunks of `--chunk` rows at successive offsets of ONE record block, in forward and in
reverse order (position vs time-in-sequence), records and output from torch's allocator like bench.py's."""
import argparse
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import pgen_rs_amd
from pgen_rs_amd import _capi
from pgen_rs_amd.synth import keep_indices


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=500000)
    ap.add_argument("--variants", type=int, default=125000)
    ap.add_argument("--chunk", type=int, default=24576)
    ap.add_argument("--keep-modulus", type=int, default=100)
    ap.add_argument("--same-out", action="store_true", help="every chunk writes the SAME output rows (isolates the record position)")
    args = ap.parse_args()
    n, v, c = args.samples, args.variants, args.chunk
    kept = keep_indices(n, modulus=args.keep_modulus)
    eng = pgen_rs_amd.GtEngine(n, kept_idx=kept, device=0)
    r, row = eng.record_size, eng.gt_row_bytes
    recs = eng.synth_records(v)
    out = torch.empty(v * row, dtype=torch.uint8, device="cuda:0")
    parts = v // c
    # an independent reader on the same regions: torch's int64 sum over each part
    for rep in range(2):
        line = []
        for q in range(parts):
            lo = (q * c * r + 7) // 8 * 8
            view = recs[lo : lo + (c * r) // 8 * 8 - 8].view(torch.int64)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            _ = view.sum()
            b.record()
            torch.cuda.synchronize()
            line.append(f"p{q}:{c * r / a.elapsed_time(b) / 1e9:.2f}")
        print("torch int64 sum, TB/s:", " ".join(line))
    print("records at", hex(recs.data_ptr()), "out at", hex(out.data_ptr()))
    res = {}
    for order in ("fwd", "rev", "fwd", "rev"):
        seq = list(range(parts)) if order == "fwd" else list(reversed(range(parts)))
        evs = []
        torch.cuda.synchronize()
        for q in seq:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            eng.decode_emit(recs, c, out=out, records_offset=q * c * r, out_offset=0 if args.same_out else q * c * row)
            b.record()
            evs.append((q, a, b))
        torch.cuda.synchronize()
        line = []
        for q, a, b in evs:
            ms = a.elapsed_time(b)
            res.setdefault(q, []).append(ms)
            line.append(f"p{q}:{ms:.3f}")
        print(order, " ".join(line))
    alg = c * (r + row)
    for q in range(parts):
        med = statistics.median(res[q])
        print(f"part {q}: median {med:.3f} ms  {alg / med / 1e9:.2f} TB/s total, read {c * r / med / 1e9:.2f} TB/s")


if __name__ == "__main__":
    main()
