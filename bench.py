#!/usr/bin/env python3
"""bench.py — the GT decode/emit hot path on MI355X, measured the way BASELINE.json asks.

    python bench.py --gpus 1 --steps K --warmup W                      # BASELINE configs[2]: 100 000 x 500 000, K = N
    python bench.py --config {c3,chr22,c4,c5,c4shard,c5shard,basic2}   # the other named shapes
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W         # BASELINE configs[3]: 1 M x 500 000, variant-sharded

A config names a TOTAL workload (V variants x N samples, a keep rule).  Its kept-variant list is cut into
`world` contiguous ranges by `pgen_rs_amd.sharding.shard_range` (the reference's outer loop,
src/pfile.rs:156, iterated in file order) and rank r decodes range r — strong scaling, no collective on the
data path.  A "step" = one pass of the hot path (src/pfile.rs:165-190) over the rank's whole
range: the range's mode-0x02 records are resident in HBM before the timed region starts; the GT text goes to
an output buffer in HBM that holds as many variants as fit beside the records (one launch per step when the
whole range fits — configs[2] does: 12.5 GB in, 200 GB out — otherwise a few launches that re-use the buffer,
as a D2H pipeline would).

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts its own ranks: the parent — which
never imports torch or touches a GPU — runs `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a
CHILD process, relays its output and returns its exit code (never a re-exec).

One JSON line on stdout (rank 0).  Besides the contract's keys:
  `roofline`       HBM; algorithmic bytes (R + 4K + 1 per variant) / hipEvent time on the
                   kernel's stream; per-step event pairs give min / median / max (the output buffer's
                   physical placement moves a launch by 5-9 % between processes);
                   `traffic` MEASURED IN THE RUN by two child passes under rocprofv3 --pmc (FETCH_SIZE, WRITE_SIZE);
                   if the profiler is not available: from the committed PMC passes of the same shape (profiles/r0*/pmc_summary.json)
  `cpu_baseline`   the oracle's literal restatement of the reference loop, 1 core, same N, bounded V
  `host_delivered` PCIe-inclusive rate (pinned host records -> H2D || kernel || D2H -> pinned host text),
                   measured OUTSIDE the timed region; it is never `value`
  `self_check`     structure of every row + a torch re-encode round trip on sampled rows (no oracle here)
  `secondary`      the other BASELINE shapes, a few steps each AFTER the headline and outside its timed region:
                   N = 1: `c5shard` (north_star's ">= 50 % of the HBM-read roofline on the 2-bit unpack" lives here:
                   `read_only_frac`), `chr22`, `basic2`;  N > 1: `c5` (configs[4]) on the ranks' resident c4 records
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import sys
import time
from pathlib import Path

REPO_ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO_ROOT))

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec (MI355X_MICROARCH.md)
PCIE_PEAK_GBS = 63.0    # PCIe Gen5 x16 per GPU (MI355X_MICROARCH.md)
SEED_DATA = 0x5047454E

# name -> (total variants, samples, keep modulus (0 = all samples kept), what it is)
CONFIGS = {
    "c3": (100_000, 500_000, 0, "BASELINE.json configs[2]: synthetic 100k variants x 500k samples, all variants all samples"),
    "chr22": (1_103_547, 2_504, 0, "BASELINE.json configs[1]: 1000G chr22 shape, all variants all samples"),
    "c4": (1_000_000, 500_000, 0, "BASELINE.json configs[3]: synthetic 1M variants x 500k samples, variant-sharded"),
    "c5": (1_000_000, 500_000, 100, "BASELINE.json configs[4]: synthetic 1M x 500k with the 1% sample-keep mask, variant-sharded"),
    "c4shard": (125_000, 500_000, 0, "one GPU's 1/8 shard of BASELINE.json configs[3] (125k variants x 500k samples)"),
    "c5shard": (125_000, 500_000, 100, "one GPU's 1/8 shard of BASELINE.json configs[4] (125k x 500k, 1% sample-keep mask)"),
    "basic2": (200_000, 300, 0, "the reference's own dataset shape (data/basic2, data/random1/info.txt: 200k variants x 300 samples)"),
}


DISTRIBUTIONS = {
    "uniform": "uniform 2-bit codes (splitmix64 counter generator, SURVEY.md §8d)",
    "hwe": "hwe: per-variant allele frequency in [0.01, 0.5), Hardy-Weinberg genotype proportions, 0.1 % missing (PGENHIP_SYNTH_HWE, SURVEY.md §8d)",
}


def cpu_baseline(n_samples: int, kept, target_s: float = 12.0) -> dict:
    """Times the oracle's file-to-file literal restatement of src/pfile.rs:149-192 on one core.

    oracle/ is test infrastructure: it is used here only as the timed CPU baseline, never on the measured
    GPU path.  The reference itself is Rust and cannot be built in this image.  Same N and keep list as the
    GPU workload; V is cut so the VCF body on tmpfs stays <= ~1.5 GB per pass.
    """
    sys.path.insert(0, str(REPO_ROOT / "oracle"))
    import numpy as np
    import pgen_oracle as oracle  # noqa: E402

    k = n_samples if kept is None else int(len(kept))
    r = oracle.variant_record_size(n_samples)
    row = 4 * k + 1
    # cost is ~ per kept genotype + per record byte read: bound both the output and the input
    sample_variants = int(max(64, min(200_000, 1.5e9 // row, 1.0e9 // max(r, 1))))
    shm = Path("/dev/shm") if Path("/dev/shm").is_dir() else Path("/tmp")
    pgen = shm / f"pgenhip_bench_{os.getpid()}.pgen"
    out = shm / f"pgenhip_bench_{os.getpid()}.vcfbody"
    outs = []
    try:
        recs = oracle.synth_records(n_samples, sample_variants, 0, SEED_DATA)
        header = bytes([0x6C, 0x1B, 0x02]) + sample_variants.to_bytes(4, "little") + n_samples.to_bytes(4, "little") + b"\x40"
        with open(pgen, "wb") as f:
            f.write(header)
            f.write(recs.tobytes())
        del recs
        passes, dt = 0, 0.0
        while dt < target_s and passes < 64:
            t0 = time.perf_counter()
            rc = oracle.output_vcf_body_file(str(pgen), n_samples, str(out), n_var=sample_variants, kept_idx=kept)
            dt += time.perf_counter() - t0
            passes += 1
            if rc != 0:
                raise RuntimeError(f"oracle baseline failed: {rc}")
        out_bytes = out.stat().st_size
        assert out_bytes == sample_variants * row
        out.unlink()
        # the same loop on all host cores (variant ranges -> separate files): NOT the reference's behaviour
        # (pgen-rs is single-threaded), reported beside it as BASELINE.md asks
        from concurrent.futures import ThreadPoolExecutor

        cores = max(1, min(os.cpu_count() or 1, 64))
        bounds = [sample_variants * i // cores for i in range(cores + 1)]
        idx = [np.arange(bounds[i], bounds[i + 1], dtype=np.uint32) for i in range(cores)]
        outs = [shm / f"pgenhip_bench_{os.getpid()}_{i}.vcfbody" for i in range(cores)]

        def one(i):  # ctypes releases the GIL for the duration of the C call
            return oracle.output_vcf_body_file(str(pgen), n_samples, str(outs[i]), var_idx=idx[i], kept_idx=kept)

        reps, t_all = 0, 0.0
        with ThreadPoolExecutor(max_workers=cores) as ex:
            while t_all < target_s / 3 and reps < 64:
                t0 = time.perf_counter()
                rcs = list(ex.map(one, range(cores)))
                t_all += time.perf_counter() - t0
                reps += 1
                if any(rcs):
                    raise RuntimeError(f"oracle baseline failed: {rcs}")
        all_cores = {"value": reps * sample_variants * n_samples / t_all, "unit": "genotypes/s", "cores": cores, "seconds": t_all,
                     "note": "same C loop, variant ranges on all host cores, one output file per thread; not the reference's behaviour"}
    finally:
        for p in [pgen, out, *outs]:
            try:
                p.unlink()
            except FileNotFoundError:
                pass
    return {
        "value": passes * sample_variants * n_samples / dt,
        "unit": "genotypes/s",
        "cores": 1,
        "kind": "port",
        "sample": f"first {sample_variants} variants x {n_samples} samples ({k} kept) of the same synthetic workload x {passes} passes, "
                  f".pgen and VCF body on tmpfs, C restatement of src/pfile.rs:149-192 "
                  f"(per-variant alloc+seek+read, two 8-KiB-BufWriter writes per genotype), {dt:.2f} s",
        "seconds": dt,
        "vcf_MB_per_s": passes * out_bytes / dt / 1e6,
        "host_cores_available": os.cpu_count(),
        "all_cores": all_cores,
    }


def host_delivered(torch, pgen_rs_amd, dev_index: int, n: int, kept, v_avail: int, target_out_bytes: float = 12e9, hwe: bool = False) -> dict:
    """PCIe-inclusive rate of the same path (src/pfile.rs:149-190 without the file system): records start in
    pinned host memory, blocks go H2D -> pgenhip_decode_emit -> D2H into a pinned host ring on two streams
    with one ctx each (what host/pfile.cpp does around its pread/pwrite).  Run OUTSIDE the timed region."""
    dev = torch.device("cuda", dev_index)
    with pgen_rs_amd.GtEngine(n, kept_idx=kept, device=dev_index) as e0, pgen_rs_amd.GtEngine(n, kept_idx=kept, device=dev_index) as e1:
        engs = (e0, e1)
        r, row = e0.record_size, e0.gt_row_bytes
        per_variant = r + row
        block = int(max(1, min(65_536, (256 << 20) // per_variant)))   # ~256 MiB over the link per block
        v = int(max(block, min(v_avail, target_out_bytes // per_variant)))
        v = min(v, v_avail)
        n_blocks = (v + block - 1) // block
        h_recs = torch.empty(v * r, dtype=torch.uint8).pin_memory()
        h_recs.copy_(e0.synth_records(v, seed=SEED_DATA, hwe=hwe)[: v * r].cpu())
        ring = 4
        h_out = [torch.empty(block * row, dtype=torch.uint8).pin_memory() for _ in range(ring)]
        streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
        d_recs = [torch.empty(block * r, dtype=torch.uint8, device=dev) for _ in range(2)]
        d_out = [torch.empty(block * row, dtype=torch.uint8, device=dev) for _ in range(2)]
        for s, e in zip(streams, engs):
            with torch.cuda.stream(s):
                e.use_torch_stream()  # binds the ctx to torch's current stream = s

        def run():
            for i in range(n_blocks):
                b0 = i * block
                nb = min(block, v - b0)
                k2 = i & 1  # stream, device buffers; ring slot i % 4 is re-used by the same stream two blocks later (stream-ordered)
                with torch.cuda.stream(streams[k2]):
                    d_recs[k2][: nb * r].copy_(h_recs[b0 * r : (b0 + nb) * r], non_blocking=True)
                    engs[k2].decode_emit(d_recs[k2], nb, out=d_out[k2])
                    h_out[i % ring][: nb * row].copy_(d_out[k2][: nb * row], non_blocking=True)
            torch.cuda.synchronize(dev)

        run()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            run()
            ts.append(time.perf_counter() - t0)
        t = min(ts)
        last = h_out[(n_blocks - 1) % ring]
        ok = True
        if row > 1:
            ok = int(last[0]) == 9 and int(last[2]) == 47
        nb_last = v - (n_blocks - 1) * block
        ok = ok and int(last[nb_last * row - 1]) == 10
        if not ok:
            raise RuntimeError("host_delivered: malformed GT text came back over the link")
        for e in engs:
            e.use_own_stream()
        return {
            "genotypes_per_s": v * n / t,
            "vcf_MB_per_s": v * row / t / 1e6,
            "d2h_GBps": v * row / t / 1e9,
            "h2d_GBps": v * r / t / 1e9,
            "link_GBps_both_ways": v * per_variant / t / 1e9,
            "pcie_peak_GBps": PCIE_PEAK_GBS,
            "d2h_frac_of_pcie": v * row / t / 1e9 / PCIE_PEAK_GBS,
            "seconds": t,
            "sample": f"{v} variants x {n} samples in blocks of {block} variants, 2 streams x (H2D, kernel, D2H), pinned host records and a "
                      f"{ring}-slot pinned text ring, best of 3 passes; file I/O excluded (the CLI adds pread/pwrite around the same pipeline)",
        }


def self_check(torch, recs, out, v_rows: int, n: int, r: int, k: int, kept, row_bytes: int) -> dict:
    """Parity of what the timed launches left in HBM, without the oracle: (1) every row's '\\n', first TAB and
    first '/' (strided over the whole buffer); (2) on sampled rows a re-encode round trip — text bytes 1 and 3
    of every genotype -> 2-bit code, compared with the codes torch unpacks from the record bits
    (src/pfile.rs:172-183 read backwards).  Full byte parity against the oracle is tests/'s job."""
    dev = out.device
    rows2d = out[: v_rows * row_bytes].view(v_rows, row_bytes)
    bad = int((rows2d[:, row_bytes - 1] != 10).sum())
    if k > 0:
        bad += int((rows2d[:, 0] != 9).sum()) + int((rows2d[:, 2] != 47).sum())
    sample_rows = sorted({0, v_rows - 1, *[(v_rows * i) // 61 for i in range(61)]})
    kept_t = None
    if kept is not None:
        kept_t = torch.as_tensor(kept.astype("int64"), device=dev)
    mismatches = 0
    for j in sample_rows:
        text = rows2d[j, : 4 * k].view(k, 4).to(torch.int32) if k else None
        if k == 0:
            continue
        a, b = text[:, 1], text[:, 3]
        # "0/0" -> 0, "0/1" -> 1, "1/1" -> 2, "./." -> 3
        code_text = torch.where(a == 46, torch.full_like(a, 3), (a - 48) + (b - 48))
        wellformed = ((text[:, 0] == 9) & (text[:, 2] == 47) & (((a == 46) & (b == 46)) | ((a >= 48) & (a <= 49) & (b >= 48) & (b <= 49) & (a <= b)))).all()
        rec = recs[j * r : (j + 1) * r].to(torch.int32)
        s = kept_t if kept_t is not None else torch.arange(n, device=dev)
        code_rec = (rec[s // 4] >> ((s % 4) * 2)) & 3
        mismatches += int((code_text != code_rec).sum()) + (0 if bool(wellformed) else 1)
    if bad or mismatches:
        raise RuntimeError(f"bench self-check failed: {bad} malformed row ends, {mismatches} genotype mismatches")
    return {"rows_structure_checked": v_rows, "rows_reencoded": len(sample_rows), "genotypes_reencoded": len(sample_rows) * k, "ok": True}


def load_traffic(v_launch: int, n: int, k: int):
    """HBM bytes per launch from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate
    runs, KiB units, FETCH_SIZE doubled per MI355X_MICROARCH.md) — only if the shape matches exactly."""
    for p in sorted((REPO_ROOT / "profiles").glob("r0*/pmc_summary.json"), reverse=True):  # the latest round's passes first
        try:
            t = json.loads(p.read_text())
        except (OSError, ValueError):
            continue
        sh = t.get("shape", {})
        if sh.get("variants_per_launch") == v_launch and sh.get("samples") == n and sh.get("kept") == k and t.get("hbm_bytes_per_launch"):
            return t["hbm_bytes_per_launch"], f"{p.relative_to(REPO_ROOT)}: {t.get('note', '')}"
    return None, None


def self_launch(argv: list[str], n_ranks: int) -> int:
    """`python bench.py --gpus N` outside torchrun: start the N ranks as a child `torch.distributed.run` job.  Called
    before torch is imported — this process never initialises a GPU (a process that has must not exec another program
    on this pool) — and it spawns, relays and returns the child's exit code; it never re-execs itself."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve()), *argv]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL between the ranks needs it on this pool
    env.setdefault("OMP_NUM_THREADS", "8")
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=None, text=True, cwd=str(REPO_ROOT))
    assert child.stdout is not None
    for ln in child.stdout:  # rank 0's JSON line (and anything else the ranks print) goes straight through
        sys.stdout.write(ln)
        sys.stdout.flush()
    return child.wait()


def time_steps(torch, dist, eng, recs, out, v: int, v_launch: int, r: int, steps: int, warmup: int, kernel: int, dev, world: int) -> dict:
    """W untimed warm-up steps, then exactly `steps` steps between barrier + synchronize on both sides.  Wall clock for
    `value`; the ctx's hipEvent pair on the kernels' stream for the roofline; one torch event pair per step for the spread."""

    def step() -> None:
        for b0 in range(0, v, v_launch):
            nb = min(v_launch, v - b0)
            eng.decode_emit(recs, nb, out=out, kernel=kernel, records_offset=b0 * r)

    for _ in range(warmup):
        step()
    torch.cuda.synchronize(dev)
    # per-step event pairs on the stream the kernels run on (the ctx is bound to torch's current stream)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    eng.timer_start()                      # hipEvent pair of the ctx, same stream, around the whole timed region
    for e0, e1 in evs:
        e0.record()
        step()
        e1.record()
    event_ms = eng.timer_stop()            # records + synchronises the stop event
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    return {"dt": dt, "event_ms": event_ms, "step_ms": [e0.elapsed_time(e1) for e0, e1 in evs]}


def launch_plan(torch, dev, v: int, row_bytes: int, share: int, max_launch_variants: int):
    """As many output rows as fit beside what is already resident, in even launches."""
    free_b, _total_b = torch.cuda.mem_get_info(dev)
    budget = max(free_b // share - (6 << 30), row_bytes)
    v_launch = max(1, min(v, budget // row_bytes)) if v else 0
    if max_launch_variants:
        v_launch = min(v_launch, max_launch_variants)
    n_launch = (v + v_launch - 1) // v_launch if v else 0
    if n_launch:
        v_launch = (v + n_launch - 1) // n_launch  # even launches
    return v_launch, n_launch


def secondary_one(torch, pgen_rs_amd, name: str, dev_index: int, steps: int, warmup: int, hwe: bool) -> dict:
    """One of the other BASELINE shapes on this GPU, a few steps, event-timed on the kernels' stream; outside the
    headline's timed region (its buffers are freed first)."""
    from pgen_rs_amd.synth import keep_indices

    dev = torch.device("cuda", dev_index)
    v, n, keep_mod, desc = CONFIGS[name]
    kept = keep_indices(n, modulus=keep_mod) if keep_mod else None
    with pgen_rs_amd.GtEngine(n, kept_idx=kept, device=dev_index) as eng:
        k, r, row_bytes = eng.kept_count, eng.record_size, eng.gt_row_bytes
        recs = eng.synth_records(v, seed=SEED_DATA, hwe=hwe)
        torch.cuda.synchronize(dev)
        v_launch, n_launch = launch_plan(torch, dev, v, row_bytes, 1, 0)
        out = torch.empty(v_launch * row_bytes, dtype=torch.uint8, device=dev)
        t = time_steps(torch, None, eng, recs, out, v, v_launch, r, steps, warmup, 0, dev, 1)
        check = self_check(torch, recs[(n_launch - 1) * v_launch * r :], out, v - (n_launch - 1) * v_launch, n, r, k, kept, row_bytes)
        ms = t["event_ms"] / steps
        alg = v * (r + 4 * k + 1)
        res = {
            "workload": f"{name}: {desc}",
            "variants": v, "samples": n, "kept_samples": k, "steps": steps, "launches_per_step": n_launch,
            "ms_per_step": ms,
            "step_ms": {"min": min(t["step_ms"]), "median": statistics.median(t["step_ms"]), "max": max(t["step_ms"])},
            "genotypes_per_s": v * n / (ms * 1e-3),
            "algorithmic_bytes_per_step": alg,
            "achieved_GBps": alg / (ms * 1e-3) / 1e9,
            "frac": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "read_only_frac": v * r / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "self_check_ok": bool(check["ok"]),
        }
        del recs, out
    torch.cuda.synchronize(dev)
    torch.cuda.empty_cache()
    return res


def measure_traffic_live(argv_shape: list[str], timeout_s: int = 300):
    """HBM bytes per pgenhip_decode_emit call of THIS workload, measured now: two child runs of this script under
    `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes: the L2's memory-side counters do not fit one pass;
    `--kernel-trace` only, as the pool requires with `--pmc`), summed over every GT kernel of a call.  Units KiB; FETCH_SIZE
    doubled (gfx950 tallies the 128-B requests of 16-B-per-lane streaming loads at 64 B: MI355X_MICROARCH.md).  Children of this
    process (it never execs itself), run after its own buffers are freed.  Returns (bytes per call, note) or raises."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    if "ROCP_TOOL_LIBRARIES" in os.environ or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        raise RuntimeError("already running under a profiler")
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not Path(rocprof).exists():
        raise RuntimeError("rocprofv3 not found")
    per_call = {}
    child_args = [sys.executable, str(Path(__file__).resolve()), *argv_shape, "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-host-delivered",
                  "--no-secondary", "--no-live-traffic"]
    env = dict(os.environ, TMPDIR="/tmp")
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix=f"pgenhip_pmc_{counter.lower()}_", dir="/tmp")
        try:
            p = subprocess.run([rocprof, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--", *child_args],
                               capture_output=True, text=True, timeout=timeout_s, cwd="/tmp", env=env)
            lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
            if p.returncode != 0 or not lines:
                raise RuntimeError(f"child under rocprofv3 --pmc {counter} failed (rc {p.returncode}): {(p.stderr or p.stdout)[-300:]}")
            child = json.loads(lines[-1])
            calls = (child["steps"] + child["warmup"]) * child["config"]["launches_per_step"]
            total = 0.0
            for f in glob.glob(os.path.join(d, "*", "*_counter_collection.csv")):
                for r in csv.DictReader(open(f)):
                    if r["Counter_Name"] == counter and "pgenhip" in r["Kernel_Name"] and "synth" not in r["Kernel_Name"]:
                        total += float(r["Counter_Value"])
            if total <= 0.0 or calls <= 0:
                raise RuntimeError(f"no {counter} samples for the GT kernels")
            per_call[counter] = total / calls
        finally:
            shutil.rmtree(d, ignore_errors=True)
    hbm = int((2.0 * per_call["FETCH_SIZE"] + per_call["WRITE_SIZE"]) * 1024)
    note = (f"measured in this run: two child passes of this workload under rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (KiB per call: "
            f"{per_call['FETCH_SIZE']:.0f} / {per_call['WRITE_SIZE']:.0f}; FETCH_SIZE doubled for 16-B/lane streaming loads per MI355X_MICROARCH.md), "
            "summed over every GT kernel of one pgenhip_decode_emit call")
    return hbm, note


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", choices=sorted(CONFIGS), default=None,
                    help="named workload (default: c3 on one GPU, c4 = 1M x 500k variant-sharded on several)")
    ap.add_argument("--variants", type=int, default=None, help="custom TOTAL variants (overrides the config's)")
    ap.add_argument("--samples", type=int, default=None, help="custom sample count (overrides the config's)")
    ap.add_argument("--keep-modulus", type=int, default=None, help="keep sample i iff splitmix64(seed^i) %% m == 0 (0 = all)")
    ap.add_argument("--distribution", choices=["uniform", "hwe"], default="uniform",
                    help="value distribution of the synthetic records (SURVEY.md §8d): uniform 2-bit codes, or per-variant allele "
                         "frequencies with Hardy-Weinberg genotype proportions and 0.1 %% missing (mostly 0/0, like real data)")
    ap.add_argument("--max-launch-variants", type=int, default=0, help="cap on variants per launch (0 = as many as fit in HBM)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-delivered", action="store_true")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not measure roofline.traffic with child rocprofv3 --pmc passes; read it from the committed profile of the same shape")
    ap.add_argument("--no-secondary", action="store_true", help="skip the `secondary` block (the other BASELINE shapes after the headline)")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, one rank per GPU) or gloo (rehearsal of the N>1 path)")
    ap.add_argument("--all-ranks-on-device0", action="store_true",
                    help="rehearsal only: every rank uses cuda:0 (with --dist-backend gloo) so the N>1 code path can run on a 1-GPU box")
    ap.add_argument("--kernel", type=int, default=0, help="PGENHIP_KERNEL_* override for A/B runs (0 = automatic, the measured default)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(sys.argv[1:], args.gpus)   # before `import torch`: the parent never touches a GPU

    import torch
    import torch.distributed as dist

    import pgen_rs_amd
    from pgen_rs_amd.sharding import shard_range

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if args.all_ranks_on_device0:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    red_dev = dev if args.dist_backend == "nccl" else torch.device("cpu")  # where the timing reduction lives
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=args.dist_backend)

    cfg_name = args.config or ("c3" if world == 1 else "c4")
    v_total, n, keep_mod, cfg_desc = CONFIGS[cfg_name]
    custom = custom_geometry = False
    if args.variants is not None and args.variants != v_total:
        v_total, custom = args.variants, True
    if args.samples is not None and args.samples != n:
        n, custom, custom_geometry = args.samples, True, True
    if args.keep_modulus is not None and args.keep_modulus != keep_mod:
        keep_mod, custom, custom_geometry = args.keep_modulus, True, True

    kept = None
    if keep_mod:
        from pgen_rs_amd.synth import keep_indices

        kept = keep_indices(n, modulus=keep_mod)

    hwe = args.distribution == "hwe"
    eng = pgen_rs_amd.GtEngine(n, kept_idx=kept, device=local_rank)
    k, r, row_bytes = eng.kept_count, eng.record_size, eng.gt_row_bytes
    begin, end = shard_range(v_total, world, rank)  # this rank's contiguous slice of the kept-variant list
    v = end - begin

    # PCIe-inclusive rate first, on its own small buffers (measured outside the timed region; done before the big
    # allocations so that it does not run beside the driver unmapping 200 GB of freed output buffer)
    hd = None
    if world == 1 and rank == 0 and not args.no_host_delivered and v > 0:
        hd = host_delivered(torch, pgen_rs_amd, local_rank, n, kept, v, hwe=hwe)
        torch.cuda.synchronize(dev)
        torch.cuda.empty_cache()

    # the rank's records, resident before the timed region; then as many output rows as fit beside them
    recs = eng.synth_records(v, first_variant=begin, seed=SEED_DATA, hwe=hwe)
    torch.cuda.synchronize(dev)
    share = world if args.all_ranks_on_device0 else 1
    v_launch, n_launch = launch_plan(torch, dev, v, row_bytes, share, args.max_launch_variants)
    out = torch.empty(max(v_launch, 1) * row_bytes, dtype=torch.uint8, device=dev)

    t = time_steps(torch, dist, eng, recs, out, v, v_launch, r, args.steps, args.warmup, args.kernel, dev, world)
    dt, event_ms, step_ms = t["dt"], t["event_ms"], t["step_ms"]
    t_all = torch.tensor([dt, event_ms, max(step_ms), -min(step_ms)], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(t_all, op=dist.ReduceOp.MAX)
    dt_max, event_ms_max = float(t_all[0]), float(t_all[1])

    check = None
    if rank == 0 and v > 0:
        last_nb = v - (n_launch - 1) * v_launch  # rows the last launch of a step left in `out`
        check = self_check(torch, recs[(n_launch - 1) * v_launch * r :], out, last_nb, n, r, k, kept, row_bytes)

    # ---- secondary: the other BASELINE shapes, after the headline and outside its timed region ------------------
    secondary = None
    if not args.no_secondary and not custom_geometry and args.kernel == 0:  # (a custom VARIANT count keeps it: tests rehearse it small)
        sec_steps = max(3, min(args.steps, 5))
        if world == 1 and cfg_name == "c3":
            # one GPU: north_star's second target (>= 50 % of the HBM-READ roofline on the unpack) is physical only where few samples are
            # kept: configs[4]'s per-GPU shard; plus configs[1]'s block and the reference's own dataset shape
            del out, recs
            torch.cuda.synchronize(dev)
            torch.cuda.empty_cache()
            secondary = {}
            for name, st in (("c5shard", sec_steps), ("chr22", sec_steps), ("basic2", 4 * sec_steps)):
                secondary[name] = secondary_one(torch, pgen_rs_amd, name, local_rank, st, 2, hwe)
            recs = out = None
        elif world > 1 and cfg_name == "c4" and v > 0:
            # several GPUs: configs[4] = the same 1 M x 500 k records with the 1 % keep mask — every rank's shard is already resident
            from pgen_rs_amd.synth import keep_indices

            del out
            torch.cuda.synchronize(dev)
            torch.cuda.empty_cache()
            kept5 = keep_indices(n, modulus=CONFIGS["c5"][2])
            with pgen_rs_amd.GtEngine(n, kept_idx=kept5, device=local_rank) as e5:
                k5, row5 = e5.kept_count, e5.gt_row_bytes
                vl5, nl5 = launch_plan(torch, dev, v, row5, share, 0)
                out5 = torch.empty(vl5 * row5, dtype=torch.uint8, device=dev)
                t5 = time_steps(torch, dist, e5, recs, out5, v, vl5, r, sec_steps, 2, 0, dev, world)
                t5_all = torch.tensor([t5["dt"], t5["event_ms"]], dtype=torch.float64, device=red_dev)
                dist.all_reduce(t5_all, op=dist.ReduceOp.MAX)
                ok5 = True
                if rank == 0:
                    ok5 = self_check(torch, recs[(nl5 - 1) * vl5 * r :], out5, v - (nl5 - 1) * vl5, n, r, k5, kept5, row5)["ok"]
                del out5
            if rank == 0:
                dt5, ev5 = float(t5_all[0]), float(t5_all[1])
                alg5 = v * (r + 4 * k5 + 1)
                secondary = {"c5": {
                    "workload": f"c5: {CONFIGS['c5'][3]}" + (f" (custom: {v_total} variants)" if custom else "") +
                                f"; rank r decodes its contiguous 1/{world} of the variants, the records the c4 headline left resident",
                    "variants_total": v_total, "samples": n, "kept_samples": k5, "steps": sec_steps, "launches_per_step": nl5,
                    "value": v_total * n * sec_steps / dt5, "unit": "genotypes/s", "ms_per_step": dt5 / sec_steps * 1e3,
                    "kernel_ms_per_step_max_over_ranks": ev5 / sec_steps,
                    "frac": alg5 / (ev5 / sec_steps * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "read_only_frac": v * r / (ev5 / sec_steps * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "self_check_ok": bool(ok5),
                }}

    if rank == 0:
        genotypes_per_step = v_total * n                       # every 2-bit code of every record is decoded
        launches = args.steps * n_launch
        alg_bytes_per_launch = v_launch * (r + 4 * k + 1)      # R + 4K + 1 per variant
        alg_bytes_per_step = v * (r + 4 * k + 1)
        launch_ms = event_ms_max / launches                    # max-over-ranks event time / launches per rank
        achieved_gbs = alg_bytes_per_step / (event_ms_max / args.steps * 1e-3) / 1e9
        traffic, traffic_note = None, None
        if world == 1 and not args.no_live_traffic:
            # live: free this process's buffers first, the children need the same HBM
            recs = out = None
            torch.cuda.synchronize(dev)
            torch.cuda.empty_cache()
            shape_argv = ["--config", cfg_name, "--distribution", args.distribution]
            if custom:
                shape_argv += ["--variants", str(v_total), "--samples", str(n), "--keep-modulus", str(keep_mod)]
            if args.max_launch_variants:
                shape_argv += ["--max-launch-variants", str(args.max_launch_variants)]
            try:
                traffic, traffic_note = measure_traffic_live(shape_argv)
            except Exception as e:  # noqa: BLE001 — the line must come out whatever the profiler does
                traffic_note = f"live PMC passes not available ({type(e).__name__}: {str(e)[:200]}); "
        if traffic is None:
            t_file, note_file = load_traffic(v_launch, n, k)
            traffic, traffic_note = t_file, (traffic_note or "") + (f"from the committed profile {note_file}" if note_file else "no committed profile of this shape")
        shape = f"{v_total} variants x {n} samples, " + ("all samples kept" if kept is None else f"{k} samples kept (splitmix64 mask, 1/{keep_mod})")
        if custom:
            workload = f"custom shape ({shape}); derived from preset {cfg_name}"
        else:
            workload = f"{cfg_name}: {cfg_desc} ({shape})"
        workload += (f"; rank r decodes its contiguous 1/{world} of the variants ({v} here)" if world > 1 else "") + \
                    f"; mode-0x02 records resident in HBM -> VCF GT text in HBM, {n_launch} launch(es) of <= {v_launch} variants per step"
        med = statistics.median(step_ms)
        line = {
            "metric": "genotypes decoded/sec (variants x samples / s)",
            "value": genotypes_per_step * args.steps / dt_max,
            "unit": "genotypes/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt_max / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {
                "workload": workload,
                "preset": "custom" if custom else cfg_name,
                "variants_total": v_total,
                "variants_this_rank": v,
                "variants_per_launch": v_launch,
                "launches_per_step": n_launch,
                "samples": n,
                "kept_samples": k,
                "sharding": f"contiguous variant ranges, {world} rank(s), no collective",
                "distribution": DISTRIBUTIONS[args.distribution],
            },
            "vcf_MB_per_s": v_total * (4 * k + 1) * args.steps / dt_max / 1e6,
            "genotypes_emitted_per_s": v_total * k * args.steps / dt_max,
            "roofline": {
                "bound": "hbm",
                "achieved": achieved_gbs,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved_gbs / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_note": traffic_note,
                "algorithmic_bytes_per_launch": alg_bytes_per_launch,
                "kernel_ms_avg": launch_ms,
                "launches_timed": launches,
                "read_only_frac": v * r / (event_ms_max / args.steps * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "step_ms": {"min": min(step_ms), "median": med, "max": max(step_ms),
                            "max_over_ranks": float(t_all[2]), "min_over_ranks": -float(t_all[3])},
                "frac_step_best": alg_bytes_per_step / (min(step_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "frac_step_median": alg_bytes_per_step / (med * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "frac_step_worst": alg_bytes_per_step / (max(step_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS,
            },
            "self_check": check,
        }
        if secondary is not None:
            line["secondary"] = secondary
        if hd is not None:
            line["host_delivered"] = hd
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(n, kept)
        print(json.dumps(line), flush=True)

    eng.close()
    if world > 1:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
