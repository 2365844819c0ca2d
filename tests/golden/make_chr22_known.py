#!/usr/bin/env python3
"""Writes tests/golden/chr22_synth_known.json: size and sha256 of the VCF that `pgen-hip filter` must produce for the synthetic
chr22-shaped triple of SURVEY.md 8(d) (1 103 547 x 2 504, seed 0x5047454E), computed WITHOUT the product: records from the
oracle's twin of the generator, body from the oracle's literal restatement of the reference's file loop
(pgo_output_vcf_body_file, src/pfile.rs:149-192), header per src/pfile.rs:139-146.  CPU only (~1 min on 8 cores).
PARITY UNPINNED by the reference (it holds no .pgen and cannot be built here): this pins the product to the oracle at full size."""
import hashlib
import json
import os
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "oracle"))
import pgen_oracle as oracle  # noqa: E402

V, N = 1_103_547, 2_504


def prefix(i: int) -> bytes:
    return b"22\t%d\tsnp%d\tA\tG\t100\tPASS\t.\tGT" % (16050000 + 7 * i, i)


def header(kept) -> bytes:
    return (b"##fileformat=VCFv4.2\n##source=pgen-rs\n##fileformat=VCFv4.2\n##source=pgen-hip synth\n"
            b"#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + b"\t".join(b"S%06d" % s for s in kept) + b"\n")


def main():
    shm = Path("/dev/shm") if Path("/dev/shm").is_dir() else Path(tempfile.gettempdir())
    d = Path(tempfile.mkdtemp(prefix="chr22_known_", dir=shm))
    pgen = d / "c.pgen"
    block = 40_000
    with open(pgen, "wb") as f:
        f.write(bytes([0x6C, 0x1B, 0x02]) + V.to_bytes(4, "little") + N.to_bytes(4, "little") + b"\x40")
        for b0 in range(0, V, block):
            f.write(oracle.synth_records(N, min(block, V - b0), first_variant=b0).tobytes())
    known = {"variants": V, "samples": N, "note": "oracle-side size and sha256 of the whole VCF (header + body); see make_chr22_known.py"}
    for name, kept in (("keep_all", None), ("keep_mask_1pct", oracle.synth_keep(N, modulus=100))):
        h = hashlib.sha256()
        hd = header(range(N) if kept is None else kept)
        h.update(hd)
        total = len(hd)

        def one(b0):
            nb = min(block, V - b0)
            out = d / f"b{b0}"
            rc = oracle.output_vcf_body_file(str(pgen), N, str(out), var_idx=np.arange(b0, b0 + nb, dtype=np.uint32), kept_idx=kept,
                                             prefixes=[prefix(i) for i in range(b0, b0 + nb)])
            assert rc == 0
            return out

        with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
            for path in ex.map(one, range(0, V, block)):   # results in order; at most a few blocks on tmpfs at once
                data = path.read_bytes()
                path.unlink()
                h.update(data)
                total += len(data)
        known[name] = {"file_bytes": total, "sha256": h.hexdigest(), "kept_samples": N if kept is None else int(len(kept))}
        print(name, known[name], flush=True)
    pgen.unlink()
    d.rmdir()
    (Path(__file__).resolve().parent / "chr22_synth_known.json").write_text(json.dumps(known, indent=1) + "\n")


if __name__ == "__main__":
    main()
