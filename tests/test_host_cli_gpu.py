"""GPU leg of the host CLI: `pgen-hip filter` end to end (metadata filter -> records staged to
HBM -> device-side line assembly -> VCF file) against the expected file built from the oracle.
BASELINE config 1 (basic1, ALT=="G") with a synthesised basic1.pgen, plus sample filters, block
splits and the default output name."""
import hashlib
import shutil
import subprocess
from pathlib import Path

import numpy as np
import pytest

import pgen_oracle as oracle
from helpers import GOLDEN, basic1_known
from ref_vcf import expected_vcf

pytestmark = pytest.mark.gpu

REPO = Path(__file__).resolve().parent.parent
CLI = REPO / "pgen_rs_amd" / "pgen-hip"


def run(*args):
    return subprocess.run([str(CLI), *args], capture_output=True, timeout=300)


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


# ---- the reference's own dataset shape: data/basic2 (200 000 variants x 300 samples) -------------------------------------
@pytest.fixture(scope="module")
def basic2(tmp_path_factory):
    """basic2.psam is the reference's own file (data/basic2/basic2.psam: per0..per299); its .pvar and .pgen are missing from
    the mount, so they are synthesised at the documented shape (data/random1/info.txt, basic2.log)."""
    d = tmp_path_factory.mktemp("basic2")
    shutil.copy(GOLDEN / "basic2" / "basic2.psam", d / "basic2.psam")
    n, v = 300, 200_000
    with open(d / "basic2.pvar", "wb") as f:
        f.write(b"##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\n")
        f.write(b"".join(b"1\t%d\tsnp%d\tA\tG\n" % (1000 + i, i) for i in range(v)))
    recs = oracle.synth_records(n, v)
    (d / "basic2.pgen").write_bytes(bytes([0x6C, 0x1B, 0x02]) + v.to_bytes(4, "little") + n.to_bytes(4, "little") + b"\x40" + recs.tobytes())
    return d / "basic2"


def test_basic2_reference_smoke_query_ten_by_ten(basic2, tmp_path):
    # the commented-out smoke test of the reference (src/main.rs:51-90): snp0..snp9 x per0..per9 on data/basic2
    out = tmp_path / "ten.vcf"
    vq = " || ".join(f'ID == "snp{i}"' for i in range(10))
    sq = " || ".join(f'IID == "per{i}"' for i in range(10))
    p = run("filter", str(basic2), "--include-var", vq, "--include-sam", sq, "-o", str(out))
    assert p.returncode == 0, p.stderr
    snps = {b"snp%d" % i for i in range(10)}
    pers = {b"per%d" % i for i in range(10)}
    want = expected_vcf(basic2, var_pred=lambda r: r[b"ID"] in snps, sam_pred=lambda r: r[b"IID"] in pers)
    got = out.read_bytes()
    assert got == want
    lines = got.split(b"\n")
    col = next(ln for ln in lines if ln.startswith(b"#CHROM"))
    assert col.endswith(b"FORMAT\t" + b"\t".join(b"per%d" % i for i in range(10)))
    body = [ln for ln in lines if ln and not ln.startswith(b"#")]
    assert len(body) == 10 and all(len(ln.split(b"\t")) == 5 + 1 + 10 for ln in body)


def test_basic2_whole_file(basic2, tmp_path):
    """All 200 000 variants x 300 samples (rows of 1 201 bytes behind ~20-byte prefixes): 245 MB of VCF, byte for byte."""
    out = tmp_path / "all.vcf"
    p = run("filter", str(basic2), "-o", str(out), "--block-mib", "64")
    assert p.returncode == 0, p.stderr
    want = expected_vcf(basic2)
    got = out.read_bytes()
    assert len(got) == len(want) and hashlib.sha256(got).digest() == hashlib.sha256(want).digest()


@pytest.mark.parametrize("block_mib,launch_mib,extra", [(1, 1, []), (4, 4, []), (4, 24, []), (2, 1024, []), (8, 40, ["--shards", "3"]), (4, 32, ["--write-threads", "3"]),
                                                         (1, 16, ["--include-sam", 'IID != "per7"', "--include-var", 'ALT == "G"'])])
def test_basic2_launches_of_several_blocks(basic2, tmp_path, block_mib, launch_mib, extra):
    """A launch covers several blocks (`--launch-mib` / `--block-mib` = 1 (245 launches), 1, 6, all, 5, 8, 16 blocks, ramping 1, 2, 4, ...): the
    blocks are staged into, and copied out of, their places in the launch's buffers one by one.  Same bytes for every split."""
    out = tmp_path / "l.vcf"
    p = run("filter", str(basic2), "-o", str(out), "--block-mib", str(block_mib), "--launch-mib", str(launch_mib), *extra)
    assert p.returncode == 0, p.stderr
    if "--include-sam" in extra:
        want = expected_vcf(basic2, var_pred=lambda r: r[b"ALT"] == b"G", sam_pred=lambda r: r[b"IID"] != b"per7")
    else:
        want = expected_vcf(basic2)
    got = out.read_bytes()
    assert len(got) == len(want) and hashlib.sha256(got).digest() == hashlib.sha256(want).digest()


@pytest.mark.parametrize("extra", [[], ["--launch-mib", "4"], ["--bgzf", "--bgzf-level", "1"], ["--shards", "2"]])
def test_write_error_in_the_middle_of_the_body_ends_the_run(basic2, tmp_path, extra):
    """The consumer thread's write fails once the output passes 16 MiB (RLIMIT_FSIZE with SIGXFSZ ignored: EFBIG) while the producer is
    staging / launching / queueing copies ahead of it: the run must end with the reference's panic status and the errno text — not hang
    in one of the pipeline's waits, not exit 0 with a short file."""
    import resource
    import signal

    def limit():
        signal.signal(signal.SIGXFSZ, signal.SIG_IGN)
        resource.setrlimit(resource.RLIMIT_FSIZE, (16 << 20, 16 << 20))

    out = tmp_path / ("e.vcf.gz" if "--bgzf" in extra else "e.vcf")
    # (BGZF at level 1 turns the 245 MB of text into ~41 MB: past the limit too)
    p = subprocess.run([str(CLI), "filter", str(basic2), "-o", str(out), "--block-mib", "4", *extra], capture_output=True, timeout=300, preexec_fn=limit)
    assert p.returncode == 101, (p.returncode, p.stderr[-500:])
    assert b"File too large" in p.stderr or b"EFBIG" in p.stderr, p.stderr[-500:]


# ---- a variable-width (mode 0x10) file through the CLI ---------------------------------------------
@pytest.fixture(scope="module")
def vw_pfile(tmp_path_factory):
    import sys

    sys.path.insert(0, str(GOLDEN))
    import make_golden_vw as writer

    d = tmp_path_factory.mktemp("vw")
    n, v = 2504, 3000
    rng = np.random.default_rng(2026)
    types = np.where(rng.random(v) < 0.8, 0, rng.integers(1, 8, size=v)).tolist()
    types[0] = 0
    recs = writer.make_records(rng, n, types)
    data, _ = writer.write_vw(n, recs, 8, 2)
    (d / "vw.pgen").write_bytes(data)
    with open(d / "vw.pvar", "wb") as f:
        f.write(b"#CHROM\tPOS\tID\tREF\tALT\tRTYPE\n")
        f.write(b"".join(b"7\t%d\tv%d\tC\tT\t%d\n" % (500 + 3 * i, i, t) for i, t in enumerate(types)))
    with open(d / "vw.psam", "wb") as f:
        f.write(b"#IID\tSEX\n" + b"".join(b"S%04d\tNA\n" % i for i in range(n)))
    return d / "vw"


def test_variable_width_file_plain_records(vw_pfile, tmp_path):
    out = tmp_path / "plain.vcf"
    p = run("filter", str(vw_pfile), "--include-var", 'RTYPE == "0"', "--block-mib", "2", "-o", str(out))
    assert p.returncode == 0, p.stderr
    assert out.read_bytes() == expected_vcf(vw_pfile, var_pred=lambda r: r[b"RTYPE"] == b"0")
    out2 = tmp_path / "plain_sub.vcf"
    p = run("filter", str(vw_pfile), "--include-var", 'RTYPE == "0" && REF == "C"', "--include-sam", 'IID != "S0007"', "-o", str(out2))
    assert p.returncode == 0, p.stderr
    assert out2.read_bytes() == expected_vcf(vw_pfile, var_pred=lambda r: r[b"RTYPE"] == b"0", sam_pred=lambda r: r[b"IID"] != b"S0007")


def test_variable_width_file_compressed_record_exits_101(vw_pfile, tmp_path):
    p = run("filter", str(vw_pfile), "-o", str(tmp_path / "x.vcf"))   # every variant, the compressed ones too
    assert p.returncode == 101 and b"stored compressed" in p.stderr
    q = run("query", str(vw_pfile), "-i", 'RTYPE != "0"', "-f", "ID")     # metadata queries never touch the records
    assert q.returncode == 0 and q.stdout.startswith(b"v")


# ---- BGZF (`.vcf.gz`) output (format parity unpinned: the reference writes none) -----------------------
def test_bgzf_output_round_trips_to_the_plain_file(basic1, tmp_path):
    """`-o x.vcf.gz`: the same bytes as the plain file once inflated (python's gzip: an independent inflater), a BGZF EOF marker
    at the end, every member <= 64 KiB; one shard, several shards (shards > 0 go through temporary files that are appended in
    order), small blocks, and the `--bgzf` flag with the default output name."""
    import gzip

    from test_bgzf import check_bgzf

    want = expected_vcf(basic1, var_pred=lambda r: r[b"ALT"] == b"G")
    for i, extra in enumerate((["--block-mib", "64"], ["--shards", "3", "--block-mib", "4"], ["--block-mib", "1", "--bgzf-level", "1", "--compress-threads", "3"])):
        out = tmp_path / f"g{i}.vcf.gz"
        p = run("filter", str(basic1), "--include-var", 'ALT=="G"', "-o", str(out), "--stats", *extra)
        assert p.returncode == 0, p.stderr
        check_bgzf(out, want)
        assert not list(tmp_path.glob("*.tmp")), "a shard's temporary file was left behind"
        import json

        stats = json.loads(p.stderr.decode().strip().splitlines()[-1])
        assert stats["file_bytes"] == out.stat().st_size and stats["header_bytes"] + stats["body_bytes"] == len(want)
        assert out.stat().st_size < len(want) // 4
    p = run("filter", str(basic1), "--include-var", 'ID == "rs8100066"', "--include-sam", 'IID == "NA20900"', "--bgzf")
    assert p.returncode == 0, p.stderr
    default = Path(str(basic1) + ".pgen-rs.vcf.gz")
    assert gzip.decompress(default.read_bytes()) == expected_vcf(basic1, var_pred=lambda r: r[b"ID"] == b"rs8100066", sam_pred=lambda r: r[b"IID"] == b"NA20900")
    default.unlink()


def test_bgzf_lines_longer_than_a_member(tmp_path):
    """20 011 samples: every body line is 80 KB, longer than a BGZF member's 65 280 bytes of input — lines straddle members."""
    from test_bgzf import check_bgzf

    pfx = tmp_path / "wide"
    assert run("synth", str(pfx), "--variants", "700", "--samples", "20011").returncode == 0
    plain, gz = tmp_path / "w.vcf", tmp_path / "w.vcf.gz"
    assert run("filter", str(pfx), "-o", str(plain)).returncode == 0
    assert plain.read_bytes() == expected_vcf(pfx)
    assert run("filter", str(pfx), "-o", str(gz), "--block-mib", "8").returncode == 0
    check_bgzf(gz, plain.read_bytes())
