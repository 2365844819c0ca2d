"""BASELINE configs[1] at its STATED size through the CLI (VERDICT r2 missing #4 / weak #3): `pgen-hip synth` writes the
1000G-chr22-shaped triple (1 103 547 variants x 2 504 samples, 0.69 GB of .pgen) on tmpfs, `pgen-hip filter` turns it into
the 11.1-GB VCF, and every byte of that file is compared — streaming, block by block — with what the oracle's literal
restatement of the reference's file loop (`pgo_output_vcf_body_file`, src/pfile.rs:149-192, per-variant seek + read, the
8-KiB BufWriter) writes for the same .pgen with the same prefixes, plus the header of src/pfile.rs:136-146.

This is the only place that runs the product host's large-file branches: file offsets past 4 GiB through `file_off` /
`pwrite`, the default 128-MiB block with ~85 blocks, `--write-threads 4`, and (few samples kept: little text per record,
so a block is bounded by its RECORD bytes) runs of >= 64 MiB read by parallel `pread`s.  The sha256 of the 11.1-GB file is
pinned in tests/golden/chr22_synth_known.json (the bytes are a pure function of the seeds of SURVEY.md 8(d))."""
import hashlib
import json
import os
import shutil
import subprocess
import tempfile
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import numpy as np
import pytest

import pgen_oracle as oracle
from helpers import GOLDEN

pytestmark = pytest.mark.gpu

REPO = Path(__file__).resolve().parent.parent
CLI = REPO / "pgen_rs_amd" / "pgen-hip"
V, N = 1_103_547, 2_504
KNOWN = json.loads((GOLDEN / "chr22_synth_known.json").read_text())


@pytest.fixture(scope="module")
def chr22():
    shm = Path("/dev/shm")
    if not shm.is_dir() or shutil.disk_usage(shm).free < (30 << 30):
        pytest.skip("needs 30 GB of free /dev/shm (0.7 GB of inputs, the 11.1-GB VCF, the oracle's blocks)")
    d = Path(tempfile.mkdtemp(prefix="pgenhip_full_", dir=shm))
    try:
        p = subprocess.run([str(CLI), "synth", str(d / "chr22"), "--variants", str(V), "--samples", str(N)], capture_output=True)
        assert p.returncode == 0, p.stderr
        assert (d / "chr22.pgen").stat().st_size == 12 + V * 626
        yield d / "chr22"
    finally:
        shutil.rmtree(d, ignore_errors=True)


def synth_prefix(i: int) -> bytes:
    # synth_pfile's pvar row joined the way src/pfile.rs:157-161 writes it: every column + '\t', then "GT"
    return b"22\t%d\tsnp%d\tA\tG\t100\tPASS\t.\tGT" % (16050000 + 7 * i, i)


def expected_header(kept_samples) -> bytes:
    # src/pfile.rs:139-146 on synth_pfile's metadata
    return (b"##fileformat=VCFv4.2\n##source=pgen-rs\n##fileformat=VCFv4.2\n##source=pgen-hip synth\n"
            b"#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + b"\t".join(b"S%06d" % s for s in kept_samples) + b"\n")


def compare_with_oracle(prefix: Path, vcf: Path, kept, block: int = 40_000, threads: int = 12) -> str:
    """Streams `vcf` against header + the oracle's file loop run block-wise; returns the sha256 of the ORACLE's side."""
    k = N if kept is None else len(kept)
    header = expected_header(range(N) if kept is None else kept)
    sha = hashlib.sha256()
    starts = list(range(0, V, block))
    tmp = vcf.parent

    def oracle_block(b0: int) -> Path:
        nb = min(block, V - b0)
        out = tmp / f"oracle_{b0}.body"
        rc = oracle.output_vcf_body_file(str(prefix) + ".pgen", N, str(out), var_idx=np.arange(b0, b0 + nb, dtype=np.uint32), kept_idx=kept,
                                         prefixes=[synth_prefix(i) for i in range(b0, b0 + nb)])
        assert rc == 0
        return out

    with open(vcf, "rb") as f, ThreadPoolExecutor(max_workers=threads) as ex:
        got = f.read(len(header))
        assert got == header, "VCF header differs"
        sha.update(header)
        pos = len(header)
        # a window of blocks in flight: the C loop releases the GIL, the comparison below runs beside it
        pending = []
        it = iter(starts)
        for b0 in it:
            pending.append((b0, ex.submit(oracle_block, b0)))
            if len(pending) < threads:
                continue
            pos = _drain_one(pending, f, sha, pos, k, block)
        while pending:
            pos = _drain_one(pending, f, sha, pos, k, block)
        assert f.read(1) == b"", "the VCF is longer than header + body"
    assert pos == vcf.stat().st_size
    return sha.hexdigest()


def _drain_one(pending, f, sha, pos, k, block):
    b0, fut = pending.pop(0)
    path = fut.result()
    want = path.read_bytes()
    path.unlink()
    nb = min(block, V - b0)
    got = f.read(len(want))
    if got != want:
        a = np.frombuffer(got, dtype=np.uint8)
        b = np.frombuffer(want, dtype=np.uint8)
        n = min(a.size, b.size)
        first = int(np.flatnonzero(a[:n] != b[:n])[0]) if n and (a[:n] != b[:n]).any() else n
        raise AssertionError(f"VCF differs from the oracle in the block of variants {b0}..{b0 + nb}: first at file offset {pos + first} "
                             f"({a.size} vs {b.size} bytes in the block)")
    sha.update(want)
    return pos + len(want)


def test_chr22_keep_all_default_blocks_vs_oracle_file_loop(chr22):
    out = chr22.parent / "all.vcf"
    p = subprocess.run([str(CLI), "filter", str(chr22), "-o", str(out), "--stats"], capture_output=True)
    assert p.returncode == 0, p.stderr
    stats = json.loads(p.stderr.decode().strip().splitlines()[-1])
    assert stats["variants_kept"] == V and stats["samples_kept"] == N
    assert out.stat().st_size == KNOWN["keep_all"]["file_bytes"] == stats["header_bytes"] + stats["body_bytes"] > (10 << 30)
    digest = compare_with_oracle(chr22, out, None)
    assert digest == KNOWN["keep_all"]["sha256"], digest
    # the same file again with four writers per block (the other pwrite branch): identical bytes
    out4 = chr22.parent / "all4.vcf"
    p = subprocess.run([str(CLI), "filter", str(chr22), "-o", str(out4), "--write-threads", "4", "--block-mib", "96"], capture_output=True)
    assert p.returncode == 0, p.stderr
    h = hashlib.sha256()
    with open(out4, "rb") as f:
        for chunk in iter(lambda: f.read(64 << 20), b""):
            h.update(chunk)
    out4.unlink()
    out.unlink()
    assert h.hexdigest() == digest


def test_chr22_keep_mask_forces_parallel_preads(chr22):
    """`--include-sam 'KEEP == "1"'` (the 1 % mask of SURVEY.md 8(d): 26 of 2 504 samples): a block is bounded by its record
    bytes, so the host reads runs of >= 64 MiB of consecutive records — the parallel-pread branch."""
    kept = oracle.synth_keep(N, modulus=100)
    assert 10 < len(kept) < 60
    out = chr22.parent / "keep.vcf"
    p = subprocess.run([str(CLI), "filter", str(chr22), "--include-sam", 'KEEP == "1"', "-o", str(out), "--read-threads", "4", "--stats"], capture_output=True)
    assert p.returncode == 0, p.stderr
    stats = json.loads(p.stderr.decode().strip().splitlines()[-1])
    assert stats["samples_kept"] == len(kept) and stats["variants_kept"] == V
    digest = compare_with_oracle(chr22, out, kept, block=120_000)
    out.unlink()
    assert digest == KNOWN["keep_mask_1pct"]["sha256"], digest
