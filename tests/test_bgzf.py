"""BGZF (`.vcf.gz`) output of the host — CPU leg.  FORMAT PARITY UNPINNED: the reference writes no
`.vcf.gz` (it git-ignores them, /root/reference/.gitignore:3, and compares against bcftools on them, README.md:170-189).
What IS pinned: the container is BGZF as the SAM spec §4.1 defines it — checked member by member here — and the payload
round-trips through an independent inflater (python's zlib / gzip) to the exact uncompressed bytes."""
import gzip
import hashlib
import json
import shutil
import struct
import subprocess
import zlib
from pathlib import Path

import numpy as np
import pytest

from helpers import GOLDEN

REPO = Path(__file__).resolve().parent.parent
CLI = REPO / "pgen_rs_amd" / "pgen-hip"
EOF_MEMBER = bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0])


def bgzf_members(data: bytes):
    """Walks a BGZF file by its BSIZE fields; yields (inflated payload) per member and checks the framing of each."""
    pos = 0
    while pos < len(data):
        assert data[pos : pos + 4] == b"\x1f\x8b\x08\x04", f"member at {pos}: not gzip / deflate / FEXTRA"
        xlen = struct.unpack_from("<H", data, pos + 10)[0]
        assert xlen == 6 and data[pos + 12 : pos + 16] == b"BC\x02\x00", "the one extra subfield is BC, 2 bytes"
        bsize = struct.unpack_from("<H", data, pos + 16)[0] + 1
        assert bsize <= 65536 and pos + bsize <= len(data)
        cdata = data[pos + 18 : pos + bsize - 8]
        crc, isize = struct.unpack_from("<II", data, pos + bsize - 8)
        payload = zlib.decompress(cdata, wbits=-15)
        assert len(payload) == isize <= 65536 and zlib.crc32(payload) == crc
        yield payload
        pos += bsize


def check_bgzf(path: Path, want: bytes):
    data = path.read_bytes()
    assert data.endswith(EOF_MEMBER), "no BGZF EOF marker"
    members = list(bgzf_members(data))
    assert members[-1] == b""
    assert all(len(m) > 0 for m in members[:-1]), "an empty member in the middle would read as EOF"
    assert b"".join(members) == want
    assert gzip.decompress(data) == want   # and as a plain multi-member gzip stream


@pytest.mark.parametrize("size,level,threads,chunk_mib", [(0, 6, 4, 1), (1, 6, 1, 1), (65279, 1, 2, 1), (65280, 6, 3, 1), (65281, 9, 8, 1),
                                                          (3_000_001, 6, 5, 1), (3_000_001, 1, 2, 2)])
def test_bgzf_writer_structure_and_round_trip(tmp_path, size, level, threads, chunk_mib):
    rng = np.random.default_rng(size + level)
    # VCF-like text with an incompressible stretch in the middle (that piece must leave as a stored deflate block)
    text = (b"22\t16050075\tsnp1\tA\tG\t100\tPASS\t.\tGT" + b"\t0/0\t0/1\t1/1\t./." * 40 + b"\n") * (size // 190 + 1)
    raw = bytearray(text[:size])
    if size > 200_000:
        raw[70_000:200_000] = rng.integers(0, 256, size=130_000, dtype=np.uint8).tobytes()
    src, dst = tmp_path / "in.txt", tmp_path / "out.gz"
    src.write_bytes(bytes(raw))
    p = subprocess.run([str(CLI), "bgzf", str(src), str(dst), "--level", str(level), "--threads", str(threads), "--chunk-mib", str(chunk_mib)], capture_output=True)
    assert p.returncode == 0, p.stderr
    check_bgzf(dst, bytes(raw))
    if size > 200_000:
        assert dst.stat().st_size < size // 2 + 140_000   # the text compresses, the random stretch does not


def test_bgzf_is_deterministic_across_thread_counts(tmp_path):
    src = tmp_path / "in.txt"
    src.write_bytes((b"1\t2\t3\tGT\t0/0\t0/1\n" * 100_000))
    outs = []
    for t in (1, 7):
        dst = tmp_path / f"o{t}.gz"
        assert subprocess.run([str(CLI), "bgzf", str(src), str(dst), "--threads", str(t)], capture_output=True).returncode == 0
        outs.append(hashlib.sha256(dst.read_bytes()).hexdigest())
    assert outs[0] == outs[1]


def test_filter_dry_run_writes_a_complete_bgzf_header(tmp_path):
    """`filter --dry-run -o x.vcf.gz` (no GPU): the VCF header as BGZF members + EOF marker; basic1's known header (the reference's
    own data/basic1 metadata, tests/golden/basic1_known.json) comes back out of it."""
    known = json.loads((GOLDEN / "basic1_known.json").read_text())
    for ext in ("pvar", "psam"):
        shutil.copy(GOLDEN / "basic1" / f"basic1.{ext}", tmp_path / f"basic1.{ext}")
    (tmp_path / "basic1.pgen").write_bytes(bytes([0x6C, 0x1B, 0x02]) + (17784).to_bytes(4, "little") + (2504).to_bytes(4, "little") + b"\x40")
    out = tmp_path / "h.vcf.gz"
    p = subprocess.run([str(CLI), "filter", str(tmp_path / "basic1"), "--dry-run", "-o", str(out)], capture_output=True)
    assert p.returncode == 0, p.stderr
    hdr = gzip.decompress(out.read_bytes())
    assert len(hdr) == known["vcf_header_bytes"] and hashlib.sha256(hdr).hexdigest() == known["vcf_header_sha256"]
    check_bgzf(out, hdr)
    # the flag instead of the suffix, and a bad level
    out2 = tmp_path / "h2.bin"
    assert subprocess.run([str(CLI), "filter", str(tmp_path / "basic1"), "--dry-run", "--bgzf", "-o", str(out2)], capture_output=True).returncode == 0
    assert gzip.decompress(out2.read_bytes()) == hdr
    assert subprocess.run([str(CLI), "filter", str(tmp_path / "basic1"), "--dry-run", "--bgzf-level", "0", "-o", str(out2)], capture_output=True).returncode == 2
