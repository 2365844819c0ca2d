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
           "-I", str(REPO / "include"), "-o", str(exe), *[str(HOST / f) for f in ("cli.cpp", "pfile.cpp", "csvlite.cpp", "expr.cpp")],
           "-L", str(REPO / "pgen_rs_amd"), "-lpgen_hip", f"-Wl,-rpath,{REPO / 'pgen_rs_amd'}"]
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
        env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0", PGENHIP_FILTER_THREADS=threads)
        p = subprocess.run([str(asan_cli), *args], capture_output=True, env=env)
        assert p.returncode == 0, p.stderr.decode()[-2000:]
        outs.append(p.stdout)
    assert outs[0] == outs[1] and b'"variants_kept": 4130' in outs[0]

    tsan = tmp_path / "pgen-hip-tsan"
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fsanitize=thread", "-pthread", "-I", str(REPO / "include"), "-o", str(tsan),
           *[str(HOST / f) for f in ("cli.cpp", "pfile.cpp", "csvlite.cpp", "expr.cpp")],
           "-L", str(REPO / "pgen_rs_amd"), "-lpgen_hip", f"-Wl,-rpath,{REPO / 'pgen_rs_amd'}"]
    p = subprocess.run(cmd, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    env = dict(os.environ, PGENHIP_FILTER_THREADS="4", TSAN_OPTIONS="halt_on_error=1")
    p = subprocess.run([str(tsan), *args], capture_output=True, env=env)
    assert p.returncode == 0 and p.stdout == outs[0], p.stderr.decode()[-3000:]
    assert b"ThreadSanitizer" not in p.stderr
