#!/usr/bin/env python3
"""Writer + fixtures for VARIABLE-WIDTH .pgen files.

The reference holds no such file (every .pgen is missing from the mount) and only VALIDATES the format's tables
(/root/reference/src/pgen.rs, dead code), so these fixtures are written from the layout that validator walks:

    bytes 0-1   6C 1B                              (src/pgen.rs:30)
    byte  2     storage mode (0x10)                 (:34)
    3-6, 7-10   variant_count, sample_count u32 LE  (:42, :47)
    byte 11     bits 0-3 record storage mode (type bits 4 if < 4 else 8; length bytes = mode % 4 + 1), bits 4-5
                allele-count bytes (0 here), bits 6-7 provisional-ref storage (must be 1)   (:52-67)
    12 ..       one u64 LE file offset per block of 65 536 variants, ascending            (:140-169)
    then, per block: packed record types (4 bits: even variant in the low nibble; rounded up to whole bytes),
                     then record lengths (length-bytes each, LE)                           (:172-258)
    then the records back to back; a type-0 record is the mode-0x02 2-bit record of ceil(N/4) bytes
                     (src/pfile.rs:172-175, :196-200); other types carry opaque payload here.

PARITY UNPINNED: nothing from the reference pins these bytes.  Expected GT text of the type-0 records comes from
the independent numpy decoder of make_golden.py.  Run in the build container:  python tests/golden/make_golden_vw.py
"""
from __future__ import annotations

import hashlib
import json
from pathlib import Path

import numpy as np

from make_golden import decode_numpy, record_size

HERE = Path(__file__).resolve().parent
OUT = HERE / "vw"
BLOCK = 1 << 16


def write_vw(n_samples: int, records: list[tuple[int, bytes]], type_bits: int, len_bytes: int, storage_mode: int = 0x10,
             block_gap: int = 0) -> tuple[bytes, dict]:
    """-> (file bytes, expected tables).  `block_gap`: unused bytes in front of every block's records (offsets stay ascending)."""
    assert type_bits in (4, 8) and 1 <= len_bytes <= 4
    v = len(records)
    n_blocks = (v + BLOCK - 1) // BLOCK
    fmt = (0 if type_bits == 4 else 4) + (len_bytes - 1) | (1 << 6)
    head = bytes([0x6C, 0x1B, storage_mode]) + v.to_bytes(4, "little") + n_samples.to_bytes(4, "little") + bytes([fmt])
    tables = bytearray()
    for b in range(n_blocks):
        blk = records[b * BLOCK : (b + 1) * BLOCK]
        if type_bits == 4:
            t = bytearray((len(blk) + 1) // 2)
            for i, (ty, _) in enumerate(blk):
                assert 0 <= ty < 16
                t[i // 2] |= ty << (4 * (i & 1))
        else:
            t = bytearray(ty for ty, _ in blk)
        tables += t
        for _, payload in blk:
            assert len(payload) < 1 << (8 * len_bytes)
            tables += len(payload).to_bytes(len_bytes, "little")
    records_offset = 12 + 8 * n_blocks + len(tables)
    body = bytearray()
    block_offsets, offs = [], []
    for b in range(n_blocks):
        body += bytes(block_gap)
        block_offsets.append(records_offset + len(body))
        for _, payload in records[b * BLOCK : (b + 1) * BLOCK]:
            offs.append(records_offset + len(body))
            body += payload
    data = head + b"".join(o.to_bytes(8, "little") for o in block_offsets) + bytes(tables) + bytes(body)
    expect = {
        "variant_count": v, "sample_count": n_samples, "record_type_bits": type_bits, "record_length_bytes": len_bytes,
        "storage_mode": storage_mode, "block_count": n_blocks, "variant_records_offset": records_offset,
        "types": [ty for ty, _ in records], "lens": [len(p) for _, p in records], "offs": offs,
    }
    return data, expect


def make_records(rng, n_samples: int, types: list[int]) -> list[tuple[int, bytes]]:
    r = record_size(n_samples)
    out = []
    for ty in types:
        if ty == 0:
            out.append((0, rng.integers(0, 256, size=r, dtype=np.uint8).tobytes()))
        else:
            out.append((ty, rng.integers(0, 256, size=int(rng.integers(0, max(2, r))), dtype=np.uint8).tobytes()))  # opaque payload
    return out


def gt_of_type0(n_samples: int, records) -> bytes:
    r = record_size(n_samples)
    rows = [np.frombuffer(p, dtype=np.uint8) for ty, p in records if ty == 0]
    if not rows:
        return b""
    return decode_numpy(np.stack(rows).reshape(len(rows), r), n_samples, None).tobytes()


def main() -> int:
    OUT.mkdir(parents=True, exist_ok=True)
    rng = np.random.Generator(np.random.PCG64(20261004))
    specs = [
        # name, N, types, type bits, length bytes, block gap
        ("all0_4bit_len1", 11, [0] * 9, 4, 1, 0),
        ("all0_8bit_len2", 301, [0] * 40, 8, 2, 0),
        ("mixed_8bit_len2", 301, [0, 0, 1, 0, 4, 0, 0, 2, 0, 0, 0x10, 0, 1, 1, 0, 0, 6, 0, 0, 0, 3, 0, 0, 5, 0, 7, 0, 0, 0, 0x48, 0], 8, 2, 5),
        ("mixed_4bit_len3_odd", 64, [0, 1, 0, 0, 4, 0, 2], 4, 3, 0),
        ("all0_4bit_len4", 5, [0] * 5, 4, 4, 3),
        ("wide_8bit_len2", 2504, [0, 0, 0, 1, 0, 0, 4, 0, 0, 0, 0, 2, 0], 8, 2, 0),
    ]
    index = {}
    for name, n, types, tb, lb, gap in specs:
        recs = make_records(rng, n, types)
        data, expect = write_vw(n, recs, tb, lb, block_gap=gap)
        (OUT / f"{name}.pgen").write_bytes(data)
        gt = gt_of_type0(n, recs)
        (OUT / f"{name}.gt").write_bytes(gt)
        expect["gt_type0_sha256"] = hashlib.sha256(gt).hexdigest()
        expect["file_sha256"] = hashlib.sha256(data).hexdigest()
        index[name] = expect
    (OUT / "index.json").write_text(json.dumps(index, indent=1, sort_keys=True) + "\n")
    return 0


if __name__ == "__main__":
    import sys

    sys.exit(main())
