#!/usr/bin/env python3
"""Interleaved in-process A/B of kernels / knobs on one shape (replaces round 1's one-off probe scripts).

    python tools/ab_probe.py --samples 300 --variants 9200000 --arms flat runs pick auto --rounds 7
    python tools/ab_probe.py --samples 2504 --variants 1103547 --keep-frac 0.1 --arms auto pick scan

Every arm runs on the SAME record and output allocation, round-robin, so the 5-9 % process-to-process
placement spread cancels.  Prints ms (median, min) and the fraction of the 8 TB/s roofline on
R + 4K + 1 bytes per variant.  An arm is `kernel[:knob=value[,knob=value]]`, e.g. `runs:runs_rows=4`.
"""
import argparse
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import pgen_rs_amd
from pgen_rs_amd import _capi

KERNELS = {"auto": _capi.KERNEL_AUTO, "rows": _capi.KERNEL_ROWS, "flat": _capi.KERNEL_FLAT, "scan": _capi.KERNEL_SCAN,
           "wide": _capi.KERNEL_WIDE, "pick": _capi.KERNEL_PICK, "runs": _capi.KERNEL_RUNS, "rowpick": _capi.KERNEL_ROWPICK}
KNOBS = {k[len("KNOB_"):].lower(): getattr(_capi, k) for k in dir(_capi) if k.startswith("KNOB_")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, required=True)
    ap.add_argument("--variants", type=int, required=True)
    ap.add_argument("--keep-frac", type=float, default=0.0, help="random kept subset of this density (0 = all samples)")
    ap.add_argument("--keep-modulus", type=int, default=0)
    ap.add_argument("--arms", nargs="+", default=["auto"])
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--reps", type=int, default=3, help="launches per timing")
    ap.add_argument("--gather", action="store_true", help="rows through a variant index list (every other record of a file twice as long)")
    ap.add_argument("--lines", type=int, default=0, metavar="PREFIX_BYTES",
                    help="full VCF body lines (pgenhip_emit_lines) with synthetic prefixes of about this many bytes instead of GT segments")
    ap.add_argument("--keep-stride", type=int, default=0, help="regular kept subset: every S-th sample")
    ap.add_argument("--align-lines", type=int, default=0, metavar="A", help="with --lines: every line starts at a multiple of A bytes (gaps between lines stay unwritten)")
    ap.add_argument("--fold-rows", type=int, default=0, metavar="W",
                    help="with --lines: line j is written where line j %% W would go, so the whole launch writes one small window again and again "
                         "(what the kernel can issue when the memory side absorbs its writes; the output is garbage)")
    args = ap.parse_args()
    n, v = args.samples, args.variants
    kept = None
    if args.keep_frac > 0:
        kept = np.sort(np.random.default_rng(1).choice(n, size=max(4, int(n * args.keep_frac)), replace=False)).astype(np.uint32)
    elif args.keep_stride:
        kept = np.arange(0, n, args.keep_stride, dtype=np.uint32)
    elif args.keep_modulus:
        from pgen_rs_amd.synth import keep_indices
        kept = keep_indices(n, modulus=args.keep_modulus)
    arms = []
    for spec in args.arms:
        name, _, knobs = spec.partition(":")
        eng = pgen_rs_amd.GtEngine(n, kept_idx=kept, device=0)
        for kv in filter(None, knobs.split(",")):
            kk, vv = kv.split("=")
            eng.tune(KNOBS[kk], int(vv))
        arms.append((spec, KERNELS[name], eng))
    e0 = arms[0][2]
    recs = e0.synth_records(2 * v if args.gather else v)
    vidx = torch.arange(0, 2 * v, 2, dtype=torch.int32, device="cuda:0") + torch.randint(0, 2, (v,), dtype=torch.int32, device="cuda:0") if args.gather else None
    lines = None
    if args.lines:
        rng = np.random.default_rng(2)
        plen = rng.integers(max(2, args.lines - 8), args.lines + 9, size=v).astype(np.int64)
        if args.fold_rows or args.align_lines:
            plen[:] = args.lines
        poff = np.concatenate([[0], np.cumsum(plen)]).astype(np.int64)
        loff = np.concatenate([[0], np.cumsum(plen + e0.gt_row_bytes)]).astype(np.int64)
        if args.align_lines:
            pitch = -(-(args.lines + e0.gt_row_bytes) // args.align_lines) * args.align_lines
            loff = np.arange(v + 1, dtype=np.int64) * pitch
        if args.fold_rows:
            loff = loff[np.arange(v + 1) % args.fold_rows].copy()
            loff[v] = loff.max() + args.lines + e0.gt_row_bytes
        blob = torch.full((int(poff[-1]) + 1,), 65, dtype=torch.uint8, device="cuda:0")
        lines = (blob, torch.from_numpy(poff).to("cuda:0"), torch.from_numpy(loff).to("cuda:0"), int(plen.max()))
        out = torch.empty(int(loff.max()) + 64, dtype=torch.uint8, device="cuda:0")
        alg = v * e0.record_size + 2 * int(poff[-1]) + v * e0.gt_row_bytes
    else:
        out = torch.empty(v * e0.gt_row_bytes, dtype=torch.uint8, device="cuda:0")
        alg = v * (e0.record_size + e0.gt_row_bytes)

    def launch(eng, kern):
        if lines is None:
            eng.decode_emit(recs, v, out=out, kernel=kern, variant_idx=vidx)
        else:
            eng.emit_lines(recs, v, lines[0], lines[1], lines[2], lines[3], out, kernel=kern, variant_idx=vidx)

    times = {spec: [] for spec, _, _ in arms}
    ref = None
    for spec, kern, eng in arms:  # warm-up + agreement of the arms (checksum of the whole output)
        try:
            launch(eng, kern)
        except pgen_rs_amd.PgenHipError as e:
            print(f"{spec}: n/a ({e})")
            times.pop(spec)
            continue
        torch.cuda.synchronize()
        digest = int(out[: out.numel() // 8 * 8].view(torch.int64).sum().item()) if out.numel() >= 8 else 0
        ref = digest if ref is None else ref
        if digest != ref and not args.fold_rows:
            print(f"!! {spec}: output differs from the first arm")
    for _ in range(args.rounds):
        for spec, kern, eng in arms:
            if spec not in times:
                continue
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(args.reps):
                launch(eng, kern)
            b.record()
            torch.cuda.synchronize()
            times[spec].append(a.elapsed_time(b) / args.reps)
    k = e0.kept_count
    print(f"N={n} V={v} K={k}: {alg/1e9:.2f} GB algorithmic per launch")
    for spec, ts in times.items():
        med = statistics.median(ts)
        print(f"  {spec:32s} median {med:8.4f} ms  min {min(ts):8.4f} ms   frac {alg/(med*1e-3)/8e12:.3f}  (best {alg/(min(ts)*1e-3)/8e12:.3f})")


if __name__ == "__main__":
    main()
