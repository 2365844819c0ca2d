#!/usr/bin/env python3
"""One big all-samples launch against the same rows as back-to-back sub-launches of `--piece-gb` of text each (round 3: short launches of the
stream kernel run faster per byte than long ones — does cutting a long call into short launches keep that?)."""
import argparse
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import pgen_rs_amd


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, required=True)
    ap.add_argument("--variants", type=int, required=True)
    ap.add_argument("--piece-gb", type=float, nargs="+", default=[1.0, 2.0, 4.0])
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--blocks", type=int, default=0, help="PGENHIP_KNOB_WIDE_BLOCKS_PER_CU (0 = the library's rule)")
    args = ap.parse_args()
    n, v = args.samples, args.variants
    eng = pgen_rs_amd.GtEngine(n, device=0)
    if args.blocks:
        from pgen_rs_amd import _capi
        eng.tune(_capi.KNOB_WIDE_BLOCKS_PER_CU, args.blocks)
    r, row = eng.record_size, eng.gt_row_bytes
    recs = eng.synth_records(v)
    out = torch.empty(v * row, dtype=torch.uint8, device="cuda:0")
    arms = {"one launch": v}
    for g in args.piece_gb:
        arms[f"pieces of {g:g} GB"] = max(1, int(g * 1e9 // row))
    times = {k: [] for k in arms}
    for rnd in range(args.rounds + 1):
        for name, piece in arms.items():
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for r0 in range(0, v, piece):
                nb = min(piece, v - r0)
                eng.decode_emit(recs, nb, out=out, records_offset=r0 * r, out_offset=r0 * row)
            b.record()
            torch.cuda.synchronize()
            if rnd:
                times[name].append(a.elapsed_time(b))
    alg = v * (r + row)
    print(f"N={n} V={v}: {alg / 1e9:.1f} GB algorithmic")
    for name, ts in times.items():
        med = statistics.median(ts)
        print(f"  {name:22s} median {med:8.3f} ms  frac {alg / (med * 1e-3) / 8e12:.3f}  (best {alg / (min(ts) * 1e-3) / 8e12:.3f})")


if __name__ == "__main__":
    main()
