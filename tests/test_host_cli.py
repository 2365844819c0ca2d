"""The C++ host (pgen-rs's query/filter surface, pgen_rs_amd/host/) — CPU leg: query output,
filter plumbing (--dry-run), csv/evalexpr subset semantics, error exits.  BASELINE config 1
("data/basic1 filter --include-var 'ALT==\"G\"' ... plumbing, no GPU") is checked against the
known answers derived from the reference's own data/basic1 files (tests/golden/basic1_known.json)."""
import hashlib
import json
import shutil
import subprocess
from pathlib import Path

import pytest

import pgen_oracle as oracle
from helpers import GOLDEN, basic1_known

REPO = Path(__file__).resolve().parent.parent
CLI = REPO / "pgen_rs_amd" / "pgen-hip"


def run(*args, cwd=None):
    return subprocess.run([str(CLI), *args], capture_output=True, cwd=cwd)


@pytest.fixture(scope="module")
def basic1(tmp_path_factory):
    """basic1.{pvar,psam} are the reference's data files (fixtures); basic1.pgen is missing from the
    mount and synthesised with the seeded generator at the logged geometry."""
    d = tmp_path_factory.mktemp("basic1")
    for ext in ("pvar", "psam"):
        shutil.copy(GOLDEN / "basic1" / f"basic1.{ext}", d / f"basic1.{ext}")
    n, v = 2504, 17784
    recs = oracle.synth_records(n, v)
    (d / "basic1.pgen").write_bytes(bytes([0x6C, 0x1B, 0x02]) + v.to_bytes(4, "little") + n.to_bytes(4, "little") + b"\x40" + recs.tobytes())
    return d / "basic1"


def test_cli_is_built():
    assert CLI.exists(), "run __graft_entry__.build()"


def test_query_variants_known_answer(basic1):
    known = basic1_known()
    p = run("query", str(basic1), "-i", 'ALT == "G"', "-f", 'CHROM + " " + POS')
    assert p.returncode == 0, p.stderr
    assert len(p.stdout) == known["query_stdout_bytes"]
    assert hashlib.sha256(p.stdout).hexdigest() == known["query_stdout_sha256"]
    assert p.stdout.split(b"\n")[0].decode() == known["query_first_line"]


def test_query_samples_and_flag_spellings(basic1):
    a = run("query", str(basic1), "-s", "--include", 'IID == "HG00096" || IID == "HG00097"', "--fstring", "IID")
    b = run("query", str(basic1), "--samples", "-iIID == \"HG00096\" || IID == \"HG00097\"", "--fstring=IID")
    assert a.returncode == 0 and a.stdout == b"HG00096\nHG00097\n" and b.stdout == a.stdout
    # dead test of the reference (src/pfile.rs:289-306): ID == "rs8100066" keeps exactly 1 row
    c = run("query", str(basic1), "-i", 'ID == "rs8100066"', "-f", "ID")
    assert c.stdout == b"rs8100066\n"


def test_filter_plumbing_config1(basic1, tmp_path):
    known = basic1_known()
    out = tmp_path / "h.vcf"
    p = run("filter", str(basic1), "--include-var", 'ALT=="G"', "--dry-run", "-o", str(out))
    assert p.returncode == 0, p.stderr
    info = json.loads(p.stdout)
    assert info["variants_kept"] == known["alt_eq_G_kept"] == 4130
    assert info["samples_kept"] == 2504
    assert info["prefix_bytes"] == known["alt_eq_G_prefix_bytes"]
    assert info["file_bytes"] == known["alt_eq_G_file_bytes"] == 42_088_203
    hdr = out.read_bytes()
    assert len(hdr) == known["vcf_header_bytes"]
    assert hashlib.sha256(hdr).hexdigest() == known["vcf_header_sha256"]


def write_meta(d: Path, pvar: bytes, psam: bytes, n=3, v=2):
    (d / "t.pvar").write_bytes(pvar)
    (d / "t.psam").write_bytes(psam)
    r = oracle.variant_record_size(n)
    (d / "t.pgen").write_bytes(bytes([0x6C, 0x1B, 0x02]) + v.to_bytes(4, "little") + n.to_bytes(4, "little") + b"\x40" + bytes(v * r))
    return d / "t"


PSAM = b"#IID\tSEX\ns0\t1\ns1\t2\ns2\t1\n"


def test_csv_subset_semantics(tmp_path):
    # quoted field with a tab and a doubled quote, CRLF line ends, an empty line, no final newline
    pvar = b'##meta\n#CHROM\tPOS\tID\r\n1\t10\t"a\tb""c"\r\n\r\n2\t20\tplain'
    pre = write_meta(tmp_path, pvar, PSAM)
    p = run("query", str(pre), "-f", 'ID + "|" + POS')
    assert p.returncode == 0, p.stderr
    assert p.stdout == b'a\tb"c|10\nplain|20\n'


def test_ragged_record_is_an_error(tmp_path):
    pre = write_meta(tmp_path, b"#CHROM\tPOS\tID\n1\t10\n", PSAM)
    p = run("query", str(pre), "-f", "POS")
    assert p.returncode == 101 and b"fields" in p.stderr


@pytest.mark.parametrize("expr,want", [
    ('POS == "10" || POS == "20" && ID == "zzz"', b"r1\n"),          # && binds tighter than ||
    ('(POS == "10" || POS == "20") && ID != "r1"', b"r2\n"),
    ('!(CHROM == "1")', b"r2\n"),
    ('CHROM + POS == "110"', b"r1\n"),                               # + binds tighter than ==
    ('POS < "15"', b"r1\n"),                                         # strings order lexicographically
    ('1 + 2 * 3 == 7 && POS != "20"', b"r1\n"),
    ('true', b"r1\nr2\n"),
    # a digit string that does not fit i64 is a Float literal in evalexpr (i64 parse first, then f64), not an error and not a
    # saturated Int: Float == Int is false, Float > Int compares numerically (evalexpr's source is absent: unpinned)
    ('99999999999999999999 == 0', b""),
    ('99999999999999999999 > 9223372036854775807 && POS == "10"', b"r1\n"),
])
def test_expression_subset(tmp_path, expr, want):
    pre = write_meta(tmp_path, b"#CHROM\tPOS\tID\n1\t10\tr1\n2\t20\tr2\n", PSAM)
    p = run("query", str(pre), "-i", expr, "-f", "ID")
    assert p.returncode == 0, p.stderr
    assert p.stdout == want


@pytest.mark.parametrize("args", [
    ["-i", 'NOPE == "1"', "-f", "ID"],      # VariableIdentifierNotFound -> unwrap panic in the reference
    ["-i", "ID", "-f", "ID"],               # not a boolean
    ["-f", 'POS == "10"'],                  # fstring not a string
    ["-i", 'POS == 10 && "x"', "-f", "ID"],  # && on a non-boolean
    ["-f", "len(ID)"],                      # functions are outside the restated subset
    # checked integer arithmetic (evalexpr reports overflow; the restatement must not wrap, saturate or trap): ADVICE r1
    ["-i", "9223372036854775807 + 1 == 0", "-f", "ID"],
    ["-i", "0 - 9223372036854775807 - 2 == 0", "-f", "ID"],
    ["-i", "3037000500 * 3037000500 == 0", "-f", "ID"],
    ["-i", "(0 - 9223372036854775807 - 1) / (0 - 1) == 0", "-f", "ID"],
    ["-i", "(0 - 9223372036854775807 - 1) % (0 - 1) == 0", "-f", "ID"],
    ["-i", "-(0 - 9223372036854775807 - 1) == 0", "-f", "ID"],
    ["-i", "1 / 0 == 0", "-f", "ID"],
])
def test_expression_errors_exit_101(tmp_path, args):
    pre = write_meta(tmp_path, b"#CHROM\tPOS\tID\n1\t10\tr1\n", PSAM)
    p = run("query", str(pre), *args)
    assert p.returncode == 101 and p.stderr


def test_header_asserts_and_usage(tmp_path):
    pre = write_meta(tmp_path, b"#CHROM\tPOS\tID\n1\t10\tr1\n", PSAM)
    raw = bytearray((tmp_path / "t.pgen").read_bytes())
    for pos, val in ((0, 0x6D), (2, 0x10), (2, 0x01), (2, 0x03), (2, 0x20), (11, 0x00)):  # src/pfile.rs:47, :53, :69
        bad = bytearray(raw)
        bad[pos] = val
        (tmp_path / "t.pgen").write_bytes(bytes(bad))
        assert run("query", str(pre), "-f", "ID").returncode == 101
    (tmp_path / "t.pgen").write_bytes(bytes(raw))
    assert run("query", str(pre)).returncode == 2           # missing --fstring (clap usage error)
    assert run("frobnicate").returncode == 2
    assert run("query", str(tmp_path / "missing"), "-f", "ID").returncode == 101
    # psam without an IID column: "IID not among the headers" (src/pfile.rs:125-126)
    (tmp_path / "t.psam").write_bytes(b"#FID\tSEX\nf\t1\n")
    p = run("filter", str(pre), "--dry-run", "-o", str(tmp_path / "o.vcf"))
    assert p.returncode == 101 and b"IID not among the headers" in p.stderr


def test_untrusted_header_counts_allocate_nothing(tmp_path):
    """a 12-byte file claiming mode 0x10 and 2^32-1 variants used to make from_prefix allocate the offset and
    type / length tables (tens of GB, zero-filled) before looking at the file size.  Now: fstat first — exit 101 at once, and
    only mode 0x10 takes the variable-width walk (0x01 .bed, 0x03 / 0x04 dosage, 0x20 / 0x21 keep the reference's refusal, :53)."""
    import resource
    import time

    pre = write_meta(tmp_path, b"#CHROM\tPOS\tID\n1\t10\tr1\n", PSAM)
    (tmp_path / "t.pgen").write_bytes(bytes([0x6C, 0x1B, 0x10]) + (0xFFFFFFFF).to_bytes(4, "little") + (2504).to_bytes(4, "little") + b"\x40")
    t0 = time.time()
    p = subprocess.run([str(CLI), "query", str(pre), "-f", "ID"], capture_output=True,
                       preexec_fn=lambda: resource.setrlimit(resource.RLIMIT_AS, (2 << 30, 2 << 30)))  # a 2-GiB address space is plenty
    assert p.returncode == 101 and b"failed to fill whole buffer" in p.stderr, p.stderr
    assert time.time() - t0 < 5
    # tables present but the records are not: the last record must end inside the file
    import numpy as np
    from helpers import GOLDEN as G
    any_vw = sorted((G / "vw").glob("*.pgen"))[0]
    data = any_vw.read_bytes()
    (tmp_path / "t.pgen").write_bytes(data[:-1])
    p = run("query", str(pre), "-f", "ID")
    assert p.returncode == 101 and b"past the end of the file" in p.stderr, p.stderr
    for mode in (0x01, 0x03, 0x04, 0x20, 0x21):
        (tmp_path / "t.pgen").write_bytes(bytes([0x6C, 0x1B, mode]) + (1).to_bytes(4, "little") + (3).to_bytes(4, "little") + b"\x40\x00")
        p = run("query", str(pre), "-f", "ID")
        assert p.returncode == 101 and b"storage_mode == 0x02" in p.stderr, (mode, p.stderr)


def test_default_output_name_and_no_gpu_is_loud(basic1, tmp_path):
    import torch

    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU")
    p = run("filter", str(basic1), "--include-var", 'ID == "rs8100066"')
    assert p.returncode == 101 and b"no" in p.stderr.lower()


def test_parallel_metadata_filter_matches_serial(tmp_path):
    """N2: the pvar walk split between threads keeps the same records with the same indices as
    the serial walk (--filter-threads 1), reports the same error for a ragged row in a late
    piece, and stays serial when the file has a quote (a quoted field may hold a line break)."""
    import os

    rows = [b"#CHROM\tPOS\tID\tREF\tALT\tINFO"]
    for i in range(150_000):
        alt = b"G" if i % 3 == 0 else (b"T" if i % 3 == 1 else b"C")
        rows.append(b"22\t%d\tsnp%d\tA\t%s\t%s" % (16050000 + 7 * i, i, alt, b"x" * (i % 23)))
    body = b"\n".join(rows) + b"\n"
    pre = write_meta(tmp_path, body, PSAM, n=3, v=150_000)

    def dry(threads, expr='ALT == "G" || POS == "16050007"'):
        extra = [] if threads is None else ["--filter-threads", str(threads)]
        return subprocess.run([str(CLI), "filter", str(pre), "--include-var", expr, "--dry-run", "-o", str(tmp_path / "o.vcf"), *extra],
                              capture_output=True)

    serial = dry(1)
    assert serial.returncode == 0, serial.stderr
    info = json.loads(serial.stdout)
    assert info["variants_kept"] == 50_001
    for threads in (None, 2, 5, 16):
        par = dry(threads)
        assert par.returncode == 0 and par.stdout == serial.stdout, threads
    # a ragged record near the end: same exit code and message from both walks
    (tmp_path / "t.pvar").write_bytes(body + b"22\t1\tbad\n")
    a, b = dry(1), dry(7)
    assert a.returncode == b.returncode == 101 and a.stderr == b.stderr and b"fields" in a.stderr
    # a quoted field with an embedded line break: one record, whatever the thread count
    (tmp_path / "t.pvar").write_bytes(body + b'22\t2\t"q\nq"\tA\tG\tz\n')
    a, b = dry(1), dry(7)
    assert a.returncode == b.returncode == 0 and a.stdout == b.stdout
    assert json.loads(a.stdout)["variants_kept"] == 50_002
