#!/usr/bin/env python3
"""Generates the committed golden fixtures under tests/golden/.

Independent of both the C oracle and the HIP kernels: a vectorised numpy decoder written
directly from the format description (sample s = bits 2*(s%4) of byte s/4, LSB first;
00 -> 0/0, 01 -> 0/1, 10 -> 1/1, 11 -> ./. — /root/reference/src/pfile.rs:172-183) plus a numpy
restatement of the synthetic generator.  Run in the build container only:

    python tests/golden/make_golden.py            # small cases + sha256 table
    python tests/golden/make_golden.py --basic1   # also the basic1 metadata known-answers
                                                  # (reads /root/reference/data/basic1/*)

Outputs (all data, no reference source text):
  cases/<name>.pgen   mode-0x02 file: 12-byte header + fixed-width records
  cases/<name>.keep   kept sample indices, one per line (absent = keep all)
  cases/<name>.gt     expected GT segments, rows packed at 4K+1 bytes
  sha256.json         sha256 of expected GT bytes for larger seeded synthetic cases
  basic1_known.json   metadata facts of data/basic1 (row counts, header sha256, sizes)
"""
from __future__ import annotations

import argparse
import hashlib
import json
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
CASES = HERE / "cases"

# text of the four codes as a (4, 4) byte table: '\t' a '/' b
GT_TABLE = np.frombuffer(b"\t0/0\t0/1\t1/1\t./.", dtype=np.uint8).reshape(4, 4)
MASK64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def record_size(n: int) -> int:
    return (2 * n + 7) // 8


def header(n_variants: int, n_samples: int) -> bytes:
    return bytes([0x6C, 0x1B, 0x02]) + int(n_variants).to_bytes(4, "little") + int(n_samples).to_bytes(4, "little") + bytes([0x40])


def decode_numpy(records: np.ndarray, n_samples: int, kept: np.ndarray | None) -> np.ndarray:
    """records: (V, R) uint8 -> (V, 4K+1) uint8 of GT text."""
    v = records.shape[0]
    shifts = np.array([0, 2, 4, 6], dtype=np.uint8)
    codes = ((records[:, :, None] >> shifts[None, None, :]) & 3).reshape(v, -1)[:, :n_samples]
    if kept is not None:
        codes = codes[:, kept]
    k = codes.shape[1]
    out = np.empty((v, 4 * k + 1), dtype=np.uint8)
    out[:, : 4 * k] = GT_TABLE[codes].reshape(v, 4 * k)
    out[:, 4 * k] = ord("\n")
    return out


def splitmix64_np(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = (x + np.uint64(0x9E3779B97F4A7C15)) & MASK64
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & MASK64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & MASK64
        return z ^ (z >> np.uint64(31))


def synth_numpy(n_samples: int, first_variant: int, n_variants: int, seed: int, dirty_pad: bool = False) -> np.ndarray:
    """(V, R) uint8: LE words splitmix64(seed + (v << 20) + word_idx), truncated to R."""
    r = record_size(n_samples)
    words = (r + 7) // 8
    with np.errstate(over="ignore"):
        vv = (np.arange(first_variant, first_variant + n_variants, dtype=np.uint64) << np.uint64(20))[:, None]
        ctr = (np.uint64(seed) + vv + np.arange(words, dtype=np.uint64)[None, :]) & MASK64
    w = splitmix64_np(ctr)
    rec = w.astype("<u8").view(np.uint8).reshape(n_variants, words * 8)[:, :r].copy()
    if not dirty_pad and n_samples % 4 and r:
        rec[:, r - 1] &= np.uint8((1 << (2 * (n_samples % 4))) - 1)
    return rec


def synth_hwe_numpy(n_samples: int, first_variant: int, n_variants: int, seed: int) -> np.ndarray:
    """(V, R) uint8 of the "hwe" value distribution (definition in include/pgen_hip.h, PGENHIP_SYNTH_HWE)."""
    r = record_size(n_samples)
    with np.errstate(over="ignore"):
        v = np.arange(first_variant, first_variant + n_variants, dtype=np.uint64)
        p16 = (np.uint64(655) + splitmix64_np(((np.uint64(seed) ^ np.uint64(0x4D4146)) + v) & MASK64) % np.uint64(32113)).astype(np.uint64)
        kv = splitmix64_np(((np.uint64(seed) ^ np.uint64(0x485745)) + v) & MASK64)
        h = splitmix64_np((kv[:, None] + np.arange(n_samples, dtype=np.uint64)[None, :]) & MASK64)
    a = (h & np.uint64(0xFFFF)) < p16[:, None]
    b = ((h >> np.uint64(16)) & np.uint64(0xFFFF)) < p16[:, None]
    code = a.astype(np.uint8) + b.astype(np.uint8)
    code[(h >> np.uint64(32)) < np.uint64(4294967)] = 3
    padded = np.zeros((n_variants, 4 * r), dtype=np.uint8)
    padded[:, :n_samples] = code
    q = padded.reshape(n_variants, r, 4)
    return (q[:, :, 0] | (q[:, :, 1] << 2) | (q[:, :, 2] << 4) | (q[:, :, 3] << 6)).astype(np.uint8)


def keep_numpy(n_samples: int, seed: int, modulus: int) -> np.ndarray:
    i = np.arange(n_samples, dtype=np.uint64)
    h = splitmix64_np(np.uint64(seed) ^ i)
    return np.nonzero(h % np.uint64(modulus) == 0)[0].astype(np.uint32)


def write_case(name: str, records: np.ndarray, n_samples: int, kept: np.ndarray | None) -> None:
    CASES.mkdir(parents=True, exist_ok=True)
    v = records.shape[0]
    (CASES / f"{name}.pgen").write_bytes(header(v, n_samples) + records.tobytes())
    if kept is not None:
        (CASES / f"{name}.keep").write_text("".join(f"{int(i)}\n" for i in kept))
    (CASES / f"{name}.gt").write_bytes(decode_numpy(records, n_samples, kept).tobytes())


def small_cases() -> None:
    rng = np.random.Generator(np.random.PCG64(20240305))
    # hand-written truth table: 0xE4 = 0b11_10_01_00
    write_case("truth_e4", np.array([[0xE4]], dtype=np.uint8), 4, None)
    # all four codes in every byte position, dirty pad bits in the last byte
    for n in (1, 2, 3, 4, 5, 6, 7, 63, 64, 65, 255, 257):
        recs = rng.integers(0, 256, size=(5, record_size(n)), dtype=np.uint8)
        write_case(f"all_n{n}", recs, n, None)
    # subsets: K=0, K=1, clustered, sparse, dense-with-holes
    n = 301
    recs = rng.integers(0, 256, size=(7, record_size(n)), dtype=np.uint8)
    write_case("sub_k0", recs, n, np.array([], dtype=np.uint32))
    write_case("sub_k1_first", recs, n, np.array([0], dtype=np.uint32))
    write_case("sub_k1_last", recs, n, np.array([n - 1], dtype=np.uint32))
    write_case("sub_cluster", recs, n, np.arange(100, 164, dtype=np.uint32))
    write_case("sub_sparse", recs, n, np.sort(rng.choice(n, size=13, replace=False)).astype(np.uint32))
    holes = np.setdiff1d(np.arange(n), rng.choice(n, size=17, replace=False)).astype(np.uint32)
    write_case("sub_holes", recs, n, holes)
    # a wider one crossing several 64-sample groups and 16-byte chunks
    n = 2504
    recs = rng.integers(0, 256, size=(3, record_size(n)), dtype=np.uint8)
    write_case("wide_all", recs, n, None)
    write_case("wide_every7", recs, n, np.arange(3, n, 7, dtype=np.uint32))


def sha_cases() -> None:
    table = {}
    specs = [
        # name, N, first_variant, V, data seed, keep (seed, modulus) or None, dirty
        ("synth_n2504_v64", 2504, 0, 64, 0x5047454E, None, False),
        ("synth_n2504_v64_from1000", 2504, 1000, 64, 0x5047454E, None, False),
        ("synth_n10007_v16_dirty", 10007, 5, 16, 0x5047454E, None, True),
        ("synth_n50000_v8", 50000, 123456, 8, 0x5047454E, None, False),
        ("synth_n50000_v8_keep100", 50000, 123456, 8, 0x5047454E, (0x4D41534B, 100), False),
        ("synth_n50001_v8_keep3", 50001, 7, 8, 0x5047454E, (0x4D41534B, 3), False),
        ("synth_n500000_v2", 500000, 34358, 2, 0x5047454E, None, False),
        ("synth_n500000_v2_keep100", 500000, 34358, 2, 0x5047454E, (0x4D41534B, 100), False),
    ]
    hwe_specs = [
        ("hwe_n2504_v64", 2504, 0, 64, 0x5047454E, None, False),
        ("hwe_n301_v50_from777", 301, 777, 50, 0x5047454E, None, False),
        ("hwe_n50001_v8_keep100", 50001, 99_999, 8, 0x5047454E, (0x4D41534B, 100), False),
        ("hwe_n500000_v2", 500000, 34359, 2, 0x5047454E, None, False),
    ]
    for name, n, first, v, seed, keep, dirty in specs + hwe_specs:
        hwe = name.startswith("hwe_")
        recs = synth_hwe_numpy(n, first, v, seed) if hwe else synth_numpy(n, first, v, seed, dirty)
        kept = keep_numpy(n, *keep) if keep else None
        gt = decode_numpy(recs, n, kept)
        table[name] = {
            "sample_count": n,
            "first_variant": first,
            "n_variants": v,
            "seed": seed,
            "keep_seed": keep[0] if keep else None,
            "keep_modulus": keep[1] if keep else None,
            "dirty_pad": dirty,
            "distribution": "hwe" if hwe else "uniform",
            "kept_count": int(kept.size) if kept is not None else n,
            "records_sha256": hashlib.sha256(recs.tobytes()).hexdigest(),
            "gt_sha256": hashlib.sha256(gt.tobytes()).hexdigest(),
            "gt_bytes": int(gt.size),
        }
    (HERE / "sha256.json").write_text(json.dumps(table, indent=1, sort_keys=True) + "\n")


def basic1_known(ref_root: Path) -> None:
    """Metadata known-answers of data/basic1, computed by plain text handling."""
    pvar = (ref_root / "data/basic1/basic1.pvar").read_bytes()
    psam = (ref_root / "data/basic1/basic1.psam").read_bytes()
    pvar_lines = pvar.split(b"\n")
    if pvar_lines[-1] == b"":
        pvar_lines.pop()
    hdr_lines = [ln for ln in pvar_lines if ln.startswith(b"##")]
    col_line = next(ln for ln in pvar_lines if ln.startswith(b"#") and not ln.startswith(b"##"))
    rows = [ln.split(b"\t") for ln in pvar_lines if not ln.startswith(b"#")]
    cols = col_line[1:].split(b"\t")
    psam_lines = psam.split(b"\n")
    if psam_lines[-1] == b"":
        psam_lines.pop()
    sam_cols = psam_lines[0][1:].split(b"\t")
    sam_rows = [ln.split(b"\t") for ln in psam_lines[1:]]
    iid = [r[sam_cols.index(b"IID")] for r in sam_rows]
    alt = cols.index(b"ALT")
    keep_g = [i for i, r in enumerate(rows) if r[alt] == b"G"]
    # VCF header exactly as src/pfile.rs:139-146 writes it (no sample filter)
    vcf_header = b"##fileformat=VCFv4.2\n##source=pgen-rs\n" + b"".join(ln + b"\n" for ln in hdr_lines)
    vcf_header += col_line.strip() + b"\tFORMAT\t" + b"\t".join(iid) + b"\n"
    # query -i 'ALT == "G"' -f 'CHROM + " " + POS' (src/pfile.rs:78-102)
    chrom, pos = cols.index(b"CHROM"), cols.index(b"POS")
    q = b"".join(rows[i][chrom] + b" " + rows[i][pos] + b"\n" for i in keep_g)
    prefix_bytes = sum(sum(len(c) + 1 for c in rows[i]) + 2 for i in keep_g)
    n = len(sam_rows)
    known = {
        "variants": len(rows),
        "samples": n,
        "pvar_columns": [c.decode() for c in cols],
        "psam_columns": [c.decode() for c in sam_cols],
        "pvar_header_lines": len(hdr_lines),
        "alt_eq_G_kept": len(keep_g),
        "alt_eq_G_first_idx": keep_g[:5],
        "query_stdout_sha256": hashlib.sha256(q).hexdigest(),
        "query_stdout_bytes": len(q),
        "query_first_line": q.split(b"\n")[0].decode(),
        "vcf_header_bytes": len(vcf_header),
        "vcf_header_sha256": hashlib.sha256(vcf_header).hexdigest(),
        "alt_eq_G_prefix_bytes": prefix_bytes,
        "alt_eq_G_file_bytes": len(vcf_header) + prefix_bytes + len(keep_g) * (4 * n + 1),
        "index_of": {
            "rs8100066": next(i for i, r in enumerate(rows) if r[2] == b"rs8100066"),
            "rs2312724": next(i for i, r in enumerate(rows) if r[2] == b"rs2312724"),
            "rs7815": next(i for i, r in enumerate(rows) if r[2] == b"rs7815"),
            "HG00096": iid.index(b"HG00096"),
            "HG00097": iid.index(b"HG00097"),
            "NA20900": iid.index(b"NA20900"),
        },
    }
    (HERE / "basic1_known.json").write_text(json.dumps(known, indent=1, sort_keys=True) + "\n")


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--basic1", action="store_true", help="also regenerate basic1_known.json from /root/reference")
    ap.add_argument("--ref-root", default="/root/reference")
    args = ap.parse_args()
    small_cases()
    sha_cases()
    if args.basic1:
        basic1_known(Path(args.ref_root))
    return 0


if __name__ == "__main__":
    sys.exit(main())
