"""Shared helpers for the parity tests (test-side only)."""
from __future__ import annotations

import json
from pathlib import Path
from typing import Optional

import numpy as np

GOLDEN = Path(__file__).resolve().parent / "golden"


def load_case(name: str):
    """-> (n_variants, n_samples, records (V,R) uint8, kept or None, expected GT bytes)."""
    raw = (GOLDEN / "cases" / f"{name}.pgen").read_bytes()
    assert raw[:3] == b"\x6c\x1b\x02" and raw[11] == 0x40
    v = int.from_bytes(raw[3:7], "little")
    n = int.from_bytes(raw[7:11], "little")
    r = (2 * n + 7) // 8
    recs = np.frombuffer(raw[12:], dtype=np.uint8).reshape(v, r)
    keep_path = GOLDEN / "cases" / f"{name}.keep"
    kept: Optional[np.ndarray] = None
    if keep_path.exists():
        txt = keep_path.read_text().split()
        kept = np.array([int(t) for t in txt], dtype=np.uint32)
    gt = np.frombuffer((GOLDEN / "cases" / f"{name}.gt").read_bytes(), dtype=np.uint8)
    return v, n, recs, kept, gt


def case_names():
    return sorted(p.stem for p in (GOLDEN / "cases").glob("*.pgen"))


def sha_table():
    return json.loads((GOLDEN / "sha256.json").read_text())


def basic1_known():
    return json.loads((GOLDEN / "basic1_known.json").read_text())
