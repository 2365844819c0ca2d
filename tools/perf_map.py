#!/usr/bin/env python3
"""Fraction-of-roofline map of the AUTO dispatch: sample count x keep density, GT segments and full lines.

One process, one `tools/ab_probe.py`-style measurement per cell (records resident, ~6-12 GB of algorithmic bytes per launch so the
256-MiB Infinity Cache does not flatter anything, median of 3 x 3 launches, torch events on the engine's stream).  Prints a
markdown table; `profiles/r02_perf_map.md` is a run of it.
"""
import statistics
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import pgen_rs_amd  # noqa: E402

SAMPLES = [100, 300, 1000, 2504, 10_000, 100_000, 500_000]
DENSITIES = [None, 0.9, 0.5, 0.1, 0.01, 0.001]          # None = all samples
TARGET_BYTES = 8e9


def measure(n, frac, lines, gather=False):
    kept = None
    if frac is not None:
        k = int(n * frac)
        if k < 1:
            return None
        kept = np.sort(np.random.default_rng(1).choice(n, size=k, replace=False)).astype(np.uint32)
    with pgen_rs_amd.GtEngine(n, kept_idx=kept, device=0) as eng:
        per_row = eng.record_size + eng.gt_row_bytes + (2 * lines if lines else 0)
        v = int(max(1000, min(TARGET_BYTES // per_row, 60_000_000)))
        recs = eng.synth_records(2 * v if gather else v)
        vidx = (torch.arange(0, 2 * v, 2, dtype=torch.int32, device="cuda:0") + torch.randint(0, 2, (v,), dtype=torch.int32, device="cuda:0")) if gather else None
        if lines:
            rng = np.random.default_rng(2)
            plen = rng.integers(max(2, lines - 8), lines + 9, size=v).astype(np.int64)
            poff = np.concatenate([[0], np.cumsum(plen)]).astype(np.int64)
            loff = np.concatenate([[0], np.cumsum(plen + eng.gt_row_bytes)]).astype(np.int64)
            blob = torch.full((int(poff[-1]) + 1,), 65, dtype=torch.uint8, device="cuda:0")
            poff_t, loff_t = torch.from_numpy(poff).to("cuda:0"), torch.from_numpy(loff).to("cuda:0")
            out = torch.empty(int(loff[-1]), dtype=torch.uint8, device="cuda:0")
            alg = v * eng.record_size + 2 * int(poff[-1]) + v * eng.gt_row_bytes

            def launch():
                eng.emit_lines(recs, v, blob, poff_t, loff_t, int(plen.max()), out, variant_idx=vidx)
        else:
            out = torch.empty(v * eng.gt_row_bytes, dtype=torch.uint8, device="cuda:0")
            alg = v * (eng.record_size + eng.gt_row_bytes)

            def launch():
                eng.decode_emit(recs, v, out=out, variant_idx=vidx)
        launch()
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(3):
                launch()
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) / 3)
        del out, recs
    return alg / (statistics.median(ts) * 1e-3) / 8e12


def table(lines, gather=False):
    head = "| N \\ kept | " + " | ".join("all" if d is None else f"{d * 100:g} %" for d in DENSITIES) + " |"
    print(head)
    print("|" + "---|" * (len(DENSITIES) + 1))
    for n in SAMPLES:
        cells = []
        for d in DENSITIES:
            f = measure(n, d, lines, gather)
            cells.append("—" if f is None else f"{f:.2f}")
            torch.cuda.empty_cache()
        print(f"| {n} | " + " | ".join(cells) + " |", flush=True)


if __name__ == "__main__":
    print("## GT segments (`pgenhip_decode_emit`, AUTO), fraction of the 8 TB/s roofline on algorithmic bytes R + 4K + 1 per row\n")
    table(0)
    print("\n## Full lines (`pgenhip_emit_lines`, AUTO, prefixes of 22-38 bytes)\n")
    table(30)
    print("\n## Full lines, prefixes of 158-174 bytes (the reference's basic1.pvar rows: 132-248)\n")
    table(166)
    print("\n## GT segments, GATHERED rows (variant index list: every other record of a file twice as long) — an API path, the CLI packs its records\n")
    table(0, gather=True)
