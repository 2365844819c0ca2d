"""CPU-only sanitizer legs (GPU ASan is not available on this pool, so the
sanitizers run on the CPU builds): the oracle under ASan+UBSan, and the C++ host (csv/expr/Pfile/
CLI code) rebuilt with ASan+UBSan and driven through `query` / `filter --dry-run`."""
import hashlib
import os
import shutil
import subprocess
from pathlib import Path

import pytest

import pgen_oracle as oracle
from helpers import GOLDEN, basic1_known

REPO = Path(__file__).resolve().parent.parent
HOST = REPO / "pgen_rs_amd" / "host"


def test_oracle_selftest_under_asan_ubsan():
    p = subprocess.run(["make", "-C", str(REPO / "oracle"), "check"], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "oracle selftest ok" in p.stdout


@pytest.fixture(scope="module")
def asan_cli(tmp_path_factory):
    d = tmp_path_factory.mktemp("asan")
    exe = d / "pgen-hip-asan"
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-pthread",
           "-I", str(REPO / "include"), "-o", str(exe), *[str(HOST / f) for f in ("cli.cpp", "pfile.cpp", "csvlite.cpp", "expr.cpp", "bgzf.cpp")],
           "-L", str(REPO / "pgen_rs_amd"), "-lpgen_hip", "-lz", f"-Wl,-rpath,{REPO / 'pgen_rs_amd'}"]
    p = subprocess.run(cmd, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    return exe


def test_host_code_under_asan_ubsan(asan_cli, tmp_path):
    for ext in ("pvar", "psam"):
        shutil.copy(GOLDEN / "basic1" / f"basic1.{ext}", tmp_path / f"basic1.{ext}")
    n, v = 2504, 17784
    (tmp_path / "basic1.pgen").write_bytes(bytes([0x6C, 0x1B, 0x02]) + v.to_bytes(4, "little") + n.to_bytes(4, "little") + b"\x40"
                                           + oracle.synth_records(n, v).tobytes())
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="print_stacktrace=1")
    known = basic1_known()
    p = subprocess.run([str(asan_cli), "query", str(tmp_path / "basic1"), "-i", 'ALT == "G"', "-f", 'CHROM + " " + POS'], capture_output=True, env=env)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    assert hashlib.sha256(p.stdout).hexdigest() == known["query_stdout_sha256"]
    p = subprocess.run([str(asan_cli), "filter", str(tmp_path / "basic1"), "--include-var", 'ALT=="G"', "--include-sam", 'IID != "HG00096"',
                        "--dry-run", "-o", str(tmp_path / "h.vcf")], capture_output=True, env=env)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    # error paths must stay clean too (exceptions, no leaks of control flow into UB)
    for args in (["query", str(tmp_path / "basic1"), "-i", "NOPE == \"x\"", "-f", "ID"], ["query", str(tmp_path / "basic1"), "-f", "len(ID)"]):
        p = subprocess.run([str(asan_cli), *args], capture_output=True, env=env)
        assert p.returncode == 101, p.stderr.decode()[-2000:]
        assert b"AddressSanitizer" not in p.stderr and b"runtime error" not in p.stderr


def test_parallel_metadata_filter_under_asan_and_tsan(asan_cli, tmp_path):
    """N2's threaded pvar walk (forced to 4 pieces on the small basic1 file): same answer as the
    serial walk under ASan+UBSan, and no data race reports from a ThreadSanitizer build."""
    for ext in ("pvar", "psam"):
        shutil.copy(GOLDEN / "basic1" / f"basic1.{ext}", tmp_path / f"basic1.{ext}")
    n, v = 2504, 17784
    (tmp_path / "basic1.pgen").write_bytes(bytes([0x6C, 0x1B, 0x02]) + v.to_bytes(4, "little") + n.to_bytes(4, "little") + b"\x40")
    args = ["filter", str(tmp_path / "basic1"), "--include-var", 'ALT=="G"', "--dry-run", "-o", str(tmp_path / "h.vcf")]
    outs = []
    for threads in ("1", "4"):
        env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0")
        p = subprocess.run([str(asan_cli), *args, "--filter-threads", threads], capture_output=True, env=env)
        assert p.returncode == 0, p.stderr.decode()[-2000:]
        outs.append(p.stdout)
    assert outs[0] == outs[1] and b'"variants_kept": 4130' in outs[0]

    tsan = tmp_path / "pgen-hip-tsan"
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fsanitize=thread", "-pthread", "-I", str(REPO / "include"), "-o", str(tsan),
           *[str(HOST / f) for f in ("cli.cpp", "pfile.cpp", "csvlite.cpp", "expr.cpp", "bgzf.cpp")],
           "-L", str(REPO / "pgen_rs_amd"), "-lpgen_hip", "-lz", f"-Wl,-rpath,{REPO / 'pgen_rs_amd'}"]
    p = subprocess.run(cmd, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1")
    p = subprocess.run([str(tsan), *args, "--filter-threads", "4"], capture_output=True, env=env)
    assert p.returncode == 0 and p.stdout == outs[0], p.stderr.decode()[-3000:]
    assert b"ThreadSanitizer" not in p.stderr
    # the BGZF writer's deflate pool (round 3) under the same ThreadSanitizer build: pieces claimed through one atomic, members appended in order
    import gzip

    src = tmp_path / "text.vcf"
    src.write_bytes((b"22\t16050075\tsnp1\tA\tG\t100\tPASS\t.\tGT" + b"\t0/0\t0/1\t1/1\t./." * 25 + b"\n") * 9000)
    p = subprocess.run([str(tsan), "bgzf", str(src), str(tmp_path / "t.gz"), "--threads", "6", "--chunk-mib", "1"], capture_output=True, env=env)
    assert p.returncode == 0 and b"ThreadSanitizer" not in p.stderr, p.stderr.decode()[-3000:]
    assert gzip.decompress((tmp_path / "t.gz").read_bytes()) == src.read_bytes()


def test_bgzf_writer_under_asan_ubsan(asan_cli, tmp_path):
    """The BGZF writer on compressible text with an incompressible stretch (stored-block path) and on an empty file, under ASan + UBSan."""
    import gzip

    import numpy as np

    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1")
    raw = bytearray((b"1\t2\t3\tGT\t0/0\t0/1\t./.\n" * 40000))
    raw[100_000:260_000] = np.random.default_rng(5).integers(0, 256, size=160_000, dtype=np.uint8).tobytes()
    for data, threads in ((bytes(raw), "5"), (b"", "2"), (b"x", "1")):
        src, dst = tmp_path / "in.bin", tmp_path / "out.gz"
        src.write_bytes(data)
        p = subprocess.run([str(asan_cli), "bgzf", str(src), str(dst), "--threads", threads, "--chunk-mib", "1", "--level", "1"], capture_output=True, env=env)
        assert p.returncode == 0, p.stderr.decode()[-2000:]
        assert b"AddressSanitizer" not in p.stderr and b"runtime error" not in p.stderr
        assert gzip.decompress(dst.read_bytes()) == data


def test_variable_width_table_walk_fuzzed_under_asan_ubsan(tmp_path):
    """The header / offset-table walk of the variable-width modes (pgenhip_vw_*) parses bytes straight from a file.  Its source
    (pgen_rs_amd/csrc/host_pure.cpp, plain C++: the same file libpgen_hip.so is built from) is compiled with ASan + UBSan
    together with a mutation fuzzer (tests/fuzz_vw.cpp): bit flips, byte overwrites and truncations of every committed fixture
    must end in a status code — no out-of-bounds access, no overflow — and a walk that succeeds must yield non-overlapping
    records behind the tables."""
    exe = tmp_path / "fuzz_vw"
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-I", str(REPO / "include"),
           "-o", str(exe), str(REPO / "tests" / "fuzz_vw.cpp"), str(REPO / "pgen_rs_amd" / "csrc" / "host_pure.cpp")]
    p = subprocess.run(cmd, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1")
    for i, fixture in enumerate(sorted((GOLDEN / "vw").glob("*.pgen"))):
        p = subprocess.run([str(exe), str(fixture), "12000", str(1000 + i)], capture_output=True, text=True, env=env)
        assert p.returncode == 0, fixture.name + "\n" + p.stdout + p.stderr[-3000:]
        assert "AddressSanitizer" not in p.stderr and "runtime error" not in p.stderr
        assert "walked clean" in p.stdout


def test_host_cli_on_variable_width_files_under_asan(asan_cli, tmp_path):
    """`query` opens the .pgen (src/pfile.rs:41) — for a variable-width file that is the whole table walk in Pfile::from_prefix —
    and `filter --dry-run` stops before the device: both under ASan + UBSan on every fixture, and on truncated copies (exit 101)."""
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="print_stacktrace=1")
    import json

    index = json.loads((GOLDEN / "vw" / "index.json").read_text())
    for name, exp in index.items():
        data = (GOLDEN / "vw" / f"{name}.pgen").read_bytes()
        pre = tmp_path / name
        pre.with_suffix(".pgen").write_bytes(data)
        pre.with_suffix(".pvar").write_bytes(b"#CHROM\tPOS\tID\n" + b"".join(b"1\t%d\tv%d\n" % (10 + i, i) for i in range(exp["variant_count"])))
        pre.with_suffix(".psam").write_bytes(b"#IID\n" + b"".join(b"s%d\n" % i for i in range(exp["sample_count"])))
        p = subprocess.run([str(asan_cli), "query", str(pre), "-i", 'ID == "v1"', "-f", "ID"], capture_output=True, env=env)
        assert p.returncode == 0 and p.stdout == b"v1\n", p.stderr.decode()[-2000:]
        p = subprocess.run([str(asan_cli), "filter", str(pre), "--dry-run", "-o", str(tmp_path / "x.vcf")], capture_output=True, env=env)
        assert p.returncode == 0 and b'"variants_kept": %d' % exp["variant_count"] in p.stdout, p.stderr.decode()[-2000:]
        for cut in (13, 12 + 8 * exp["block_count"], exp["variant_records_offset"] - 1):
            pre.with_suffix(".pgen").write_bytes(data[:cut])
            p = subprocess.run([str(asan_cli), "query", str(pre), "-f", "ID"], capture_output=True, env=env)
            assert p.returncode == 101, (name, cut)
            assert b"AddressSanitizer" not in p.stderr and b"runtime error" not in p.stderr
