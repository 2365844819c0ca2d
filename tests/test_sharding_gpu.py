"""GPU leg of the multi-rank path: the HIP engine under
`torch.distributed` with N > 1 ranks, started exactly the way the driver starts bench.py — fresh child
processes of `python -m torch.distributed.run` — with both ranks on device 0 (this box has one GPU) and gloo
for the (tiny) control traffic.  The rank-ordered concatenation of the ranks' GT segments must be byte-equal
to the oracle's single-shard output."""
import json
import socket
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

import pgen_oracle as oracle
import pgen_rs_amd

pytestmark = pytest.mark.gpu

REPO = Path(__file__).resolve().parent.parent
CLI = REPO / "pgen_rs_amd" / "pgen-hip"


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _torchrun(nproc: int, script: Path, *args: str, timeout: int = 600) -> subprocess.CompletedProcess:
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(script), *args]
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=str(REPO))


@pytest.mark.parametrize("world,n,v,keep_mod", [(2, 2504, 4001, 0), (2, 300, 10_007, 0), (3, 40_000, 301, 100), (2, 2504, 1001, 7)])
def test_ranks_decode_their_shards_with_the_hip_engine(tmp_path, world, n, v, keep_mod):
    p = _torchrun(world, REPO / "tests" / "dist_worker.py", "--out-dir", str(tmp_path), "--samples", str(n), "--variants", str(v),
                  "--keep-modulus", str(keep_mod), "--all-ranks-on-device0")
    assert p.returncode == 0, p.stderr[-3000:]
    kept = oracle.synth_keep(n, modulus=keep_mod) if keep_mod else None
    want = oracle.decode_emit(oracle.synth_records(n, v), v, n, kept_idx=kept).tobytes()
    whole = bytearray(len(want))
    covered = 0
    for r in range(world):
        part = (tmp_path / f"part{r}.bin").read_bytes()
        off = int((tmp_path / f"part{r}.off").read_text())
        whole[off : off + len(part)] = part
        covered += len(part)
    assert covered == len(want)
    assert bytes(whole) == want


def test_bench_multi_rank_path_rehearsal():
    """bench.py's N > 1 code path (strong-scaled shards of one workload, barrier, MAX-reduce of the timings, one JSON
    line on rank 0) with 2 ranks on device 0 over gloo, on a small custom shape."""
    p = _torchrun(2, REPO / "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--dist-backend", "gloo", "--all-ranks-on-device0",
                  "--config", "chr22", "--variants", "200001", "--no-cpu-baseline", "--no-host-delivered")
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["steps"] == 3
    assert line["config"]["variants_total"] == 200_001 and line["config"]["variants_this_rank"] == 100_001
    assert line["config"]["preset"] == "custom" and "custom shape" in line["config"]["workload"]
    assert line["self_check"]["ok"] and line["value"] > 0
    # value counts the WHOLE job's genotypes over the max-over-ranks wall time
    assert abs(line["value"] - 200_001 * 2504 * 3 / (line["ms_per_step"] * 3e-3)) / line["value"] < 1e-6


def test_bench_plain_invocation_starts_its_own_ranks():
    """`python bench.py --gpus 2` the way the driver starts N = 1 — no torchrun, no WORLD_SIZE: the parent (which never
    touches a GPU) runs the torch.distributed.run job as a child, relays rank 0's line and returns its exit code.  On
    the c4 preset with a small variant count, so the N > 1 `secondary.c5` leg (configs[4] on the resident records) runs too."""
    import os

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, str(REPO / "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--dist-backend", "gloo",
                        "--all-ranks-on-device0", "--config", "c4", "--variants", "4001"], capture_output=True, text=True, timeout=900, cwd=str(REPO), env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["variants_total"] == 4001 and line["config"]["variants_this_rank"] == 2001
    assert line["config"]["samples"] == 500_000 and line["self_check"]["ok"]
    c5 = line["secondary"]["c5"]
    assert 4000 < c5["kept_samples"] < 6000  # the splitmix64 1 % mask
    assert c5["self_check_ok"] and c5["value"] > 0 and 0 < c5["read_only_frac"] < 1
    # a failing child is a failing parent (bad flag -> argparse exit 2 in every rank)
    bad = subprocess.run([sys.executable, str(REPO / "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--all-ranks-on-device0", "--config", "nope"],
                         capture_output=True, text=True, timeout=300, cwd=str(REPO), env=env)
    assert bad.returncode != 0


def test_bench_single_gpu_line_carries_the_secondary_shapes():
    """N = 1 on the c3 preset (small variant count): `secondary` holds configs[4]'s per-GPU shard (north_star's HBM-read target),
    the chr22 block and the reference's own dataset shape, each self-checked, measured after the headline."""
    p = subprocess.run([sys.executable, str(REPO / "bench.py"), "--steps", "2", "--warmup", "1", "--variants", "3001", "--no-cpu-baseline", "--no-host-delivered"],
                       capture_output=True, text=True, timeout=900, cwd=str(REPO))
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    # roofline.traffic is measured IN the run (two child passes under rocprofv3 --pmc) when the profiler is there: within a few per cent
    # of the algorithmic bytes for this write-dominated launch (it used to be read from a committed file)
    import shutil

    roof = line["roofline"]
    if shutil.which("rocprofv3") or Path("/opt/rocm/bin/rocprofv3").exists():
        assert "measured in this run" in roof["traffic_note"], roof["traffic_note"]
        assert 0.97 < roof["traffic"] / roof["algorithmic_bytes_per_launch"] < 1.05
    sec = line["secondary"]
    assert set(sec) == {"c5shard", "chr22", "basic2"}
    for name, s in sec.items():
        assert s["self_check_ok"] and s["ms_per_step"] > 0 and 0 < s["frac"] < 1, name
    assert sec["c5shard"]["variants"] == 125_000 and sec["c5shard"]["samples"] == 500_000 and 4000 < sec["c5shard"]["kept_samples"] < 6000
    assert sec["c5shard"]["read_only_frac"] > 0.3   # the target is 0.5; this only guards against a broken measurement


def test_cli_two_gpus_two_shards(tmp_path):
    """`pgen-hip filter --gpus 2 --shards 2`: per-device worker threads with their own contexts; needs 2 devices."""
    if pgen_rs_amd.device_count() < 2:
        pytest.skip("needs >= 2 HIP devices")
    pfx = tmp_path / "two"
    assert subprocess.run([str(CLI), "synth", str(pfx), "--variants", "5000", "--samples", "2504"], capture_output=True).returncode == 0
    one, two = tmp_path / "one.vcf", tmp_path / "two.vcf"
    assert subprocess.run([str(CLI), "filter", str(pfx), "-o", str(one)], capture_output=True).returncode == 0
    p = subprocess.run([str(CLI), "filter", str(pfx), "--gpus", "2", "--shards", "2", "--block-mib", "4", "-o", str(two)], capture_output=True)
    assert p.returncode == 0, p.stderr
    assert one.read_bytes() == two.read_bytes()
