"""Test-side expected-output builder for whole VCF files: a plain-Python
restatement of Pfile::output_vcf's text handling (src/pfile.rs:110-161) for metadata inside the
unambiguous subset (no quotes, \\n line ends), with the GT segments taken from the C oracle."""
from __future__ import annotations

from pathlib import Path
from typing import Callable, Optional

import numpy as np

import pgen_oracle as oracle


def read_meta(path: Path):
    """-> (leading '##' lines joined, column-header line incl. '#', column names, rows of fields)."""
    lines = path.read_bytes().split(b"\n")
    if lines and lines[-1] == b"":
        lines.pop()
    hdr = [ln for ln in lines if ln.startswith(b"#")]
    rows = [ln.split(b"\t") for ln in lines if not ln.startswith(b"#")]
    col_line = hdr[-1]
    cols = col_line[1:].split(b"\t")
    return b"".join(ln + b"\n" for ln in hdr[:-1]), col_line, cols, rows


def expected_vcf(prefix: Path, var_pred: Optional[Callable[[dict], bool]] = None,
                 sam_pred: Optional[Callable[[dict], bool]] = None) -> bytes:
    pvar_hdr, pvar_col_line, pvar_cols, pvar_rows = read_meta(prefix.with_suffix(".pvar"))
    _, _, psam_cols, psam_rows = read_meta(prefix.with_suffix(".psam"))
    raw = prefix.with_suffix(".pgen").read_bytes()
    vw_offs = None
    if raw[2] == 0x02:
        rc, v, n = oracle.parse_header(raw[:12])
        assert rc == 0
        recs = np.frombuffer(raw[12:], dtype=np.uint8)
    else:
        # variable-width file: per-variant byte offsets from the oracle's walk of the tables (src/pgen.rs)
        rc, h = oracle.vw_parse_header(raw[:12])
        assert rc == 0
        rc, _types, _lens, vw_offs = oracle.vw_index(h, raw)
        assert rc == 0
        n = int(h.sample_count)
    r = oracle.variant_record_size(n)
    keep_v = [i for i, row in enumerate(pvar_rows) if var_pred is None or var_pred(dict(zip(pvar_cols, row)))]
    keep_s = [i for i, row in enumerate(psam_rows) if sam_pred is None or sam_pred(dict(zip(psam_cols, row)))]
    iid = psam_cols.index(b"IID")
    out = [b"##fileformat=VCFv4.2\n##source=pgen-rs\n", pvar_hdr, pvar_col_line.strip(), b"\tFORMAT\t",
           b"\t".join(psam_rows[i][iid] for i in keep_s), b"\n"]
    kept = None if len(keep_s) == n else np.array(keep_s, dtype=np.uint32)
    if keep_v:
        if vw_offs is not None:
            gt = oracle.decode_emit_at(np.frombuffer(raw, dtype=np.uint8), vw_offs[keep_v], n, kept_idx=kept)
        else:
            gt = oracle.decode_emit(recs, len(keep_v), n, kept_idx=kept, record_stride=r, variant_idx=keep_v)
        row = 4 * len(keep_s) + 1
        for j, vi in enumerate(keep_v):
            out.append(b"".join(c + b"\t" for c in pvar_rows[vi]) + b"GT")
            out.append(gt[j * row : (j + 1) * row].tobytes())
    return b"".join(out)
