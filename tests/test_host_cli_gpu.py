"""GPU leg of the host CLI: `pgen-hip filter` end to end (metadata filter -> records staged to
HBM -> device-side line assembly -> VCF file) against the expected file built from the oracle.
BASELINE config 1 (basic1, ALT=="G") with a synthesised basic1.pgen, plus sample filters, block
splits and the default output name."""
import hashlib
import shutil
import subprocess
from pathlib import Path

import pytest

import pgen_oracle as oracle
from helpers import GOLDEN, basic1_known
from ref_vcf import expected_vcf

pytestmark = pytest.mark.gpu

REPO = Path(__file__).resolve().parent.parent
CLI = REPO / "pgen_rs_amd" / "pgen-hip"


def run(*args):
    return subprocess.run([str(CLI), *args], capture_output=True)


@pytest.fixture(scope="module")
def basic1(tmp_path_factory):
    d = tmp_path_factory.mktemp("basic1")
    for ext in ("pvar", "psam"):
        shutil.copy(GOLDEN / "basic1" / f"basic1.{ext}", d / f"basic1.{ext}")
    n, v = 2504, 17784
    recs = oracle.synth_records(n, v)
    (d / "basic1.pgen").write_bytes(bytes([0x6C, 0x1B, 0x02]) + v.to_bytes(4, "little") + n.to_bytes(4, "little") + b"\x40" + recs.tobytes())
    return d / "basic1"


def test_config1_filter_alt_eq_g(basic1, tmp_path):
    known = basic1_known()
    out = tmp_path / "g.vcf"
    p = run("filter", str(basic1), "--include-var", 'ALT=="G"', "-o", str(out), "--stats")
    assert p.returncode == 0, p.stderr
    got = out.read_bytes()
    assert len(got) == known["alt_eq_G_file_bytes"] == 42_088_203
    assert hashlib.sha256(got[: known["vcf_header_bytes"]]).hexdigest() == known["vcf_header_sha256"]
    want = expected_vcf(basic1, var_pred=lambda r: r[b"ALT"] == b"G")
    assert got == want


def test_small_blocks_and_sample_filter(basic1, tmp_path):
    out = tmp_path / "s.vcf"
    p = run("filter", str(basic1), "--include-var", 'REF == "A" && ALT == "C"', "--include-sam", 'IID != "HG00097" && IID != "NA20900"',
            "--block-mib", "1", "-o", str(out))
    assert p.returncode == 0, p.stderr
    want = expected_vcf(basic1, var_pred=lambda r: r[b"REF"] == b"A" and r[b"ALT"] == b"C",
                        sam_pred=lambda r: r[b"IID"] not in (b"HG00097", b"NA20900"))
    assert out.read_bytes() == want


def test_reference_smoke_query_two_by_two(basic1, tmp_path):
    # the commented-out smoke test of the reference (src/main.rs:48): rs2312724, rs7815 x HG00096, HG00097
    out = tmp_path / "two.vcf"
    p = run("filter", str(basic1), "--include-sam", 'IID == "HG00096" || IID == "HG00097"',
            "--include-var", 'ID == "rs2312724" || ID == "rs7815"', "-o", str(out))
    assert p.returncode == 0, p.stderr
    want = expected_vcf(basic1, var_pred=lambda r: r[b"ID"] in (b"rs2312724", b"rs7815"),
                        sam_pred=lambda r: r[b"IID"] in (b"HG00096", b"HG00097"))
    got = out.read_bytes()
    assert got == want
    body = got.split(b"\n")[-3:-1]
    assert body[0].startswith(b"19\t266034\trs2312724\t") and body[1].startswith(b"19\t") and len(body[0].split(b"\t")) == 8 + 1 + 2


def test_sample_filter_that_keeps_nobody(basic1, tmp_path):
    """`--include-sam` keeping ZERO samples of a non-empty .pgen is an empty kept list, not
    "all samples": every body line is the pvar columns + "GT" + '\n' (src/pfile.rs:171-190 with an empty
    inner loop) and the header line ends in "FORMAT\t" + "" (:130-146)."""
    out = tmp_path / "nobody.vcf"
    p = run("filter", str(basic1), "--include-sam", 'IID == "nobody"', "--include-var", 'ALT=="G"', "--block-mib", "1", "-o", str(out))
    assert p.returncode == 0, p.stderr
    want = expected_vcf(basic1, var_pred=lambda r: r[b"ALT"] == b"G", sam_pred=lambda r: False)
    got = out.read_bytes()
    assert len(got) == len(want) and got == want
    assert got.split(b"\n")[-2].endswith(b"\tGT")


def test_no_variants_kept_and_default_output_name(basic1):
    p = run("filter", str(basic1), "--include-var", 'ID == "nothing"')
    assert p.returncode == 0, p.stderr
    default = Path(str(basic1) + ".pgen-rs.vcf")  # src/main.rs:121-122
    assert default.read_bytes() == expected_vcf(basic1, var_pred=lambda r: False)
    default.unlink()


def test_all_variants_all_samples(basic1, tmp_path):
    out = tmp_path / "all.vcf"
    p = run("filter", str(basic1), "-o", str(out), "--block-mib", "64")
    assert p.returncode == 0, p.stderr
    assert out.read_bytes() == expected_vcf(basic1)


@pytest.mark.parametrize("shards", [2, 3, 8])
def test_logical_shards_concatenate_to_the_single_shard_file(basic1, tmp_path, shards):
    """the multi-GPU partitioner with 2/3/8 logical shards (dealt over the GPUs that
    exist — one here) must produce the byte-identical file: every range lands at its precomputed offset."""
    out = tmp_path / f"s{shards}.vcf"
    p = run("filter", str(basic1), "--include-var", 'ALT=="G"', "--include-sam", 'SEX == "NA"', "--shards", str(shards),
            "--block-mib", "4", "-o", str(out))
    assert p.returncode == 0, p.stderr
    assert out.read_bytes() == expected_vcf(basic1, var_pred=lambda r: r[b"ALT"] == b"G")
