#!/usr/bin/env python3
"""bench.py — the GT decode/emit hot path on MI355X, measured the way BASELINE.json asks.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path (src/pfile.rs:165-190) over one variant block of synthetic
.pgen records that are already resident in HBM.  Workload = BASELINE.json configs[1], the
chr22 shape (1 103 547 variants x 2 504 samples, all samples kept): 0.69 GB of packed 2-bit
records in, 11.05 GB of GT text out per step and per GPU.  Variant blocks shard with no
collective: with N GPUs every rank decodes its own block of that shape (weak
scaling), and `value` is total genotypes / max-over-ranks time.

One JSON line on stdout (rank 0).  Extra objects: `roofline` (HBM, algorithmic bytes
R + 4K + 1 per variant / hipEvent time on the kernel's stream) and `cpu_baseline` (the oracle's
literal restatement of the reference loop, 1 core, bounded sample, rank 0 at N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

REPO_ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO_ROOT))

CHR22_VARIANTS = 1_103_547
CHR22_SAMPLES = 2_504
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md)
SEED_DATA = 0x5047454E


def cpu_baseline(sample_variants: int, n_samples: int, target_s: float = 12.0) -> dict:
    """Times the oracle's file-to-file literal restatement of src/pfile.rs:149-192 on one core.

    oracle/ is test infrastructure: it is used here only as the timed CPU baseline, never on
    the measured GPU path.  The reference itself is Rust and cannot be built in this image.
    """
    sys.path.insert(0, str(REPO_ROOT / "oracle"))
    import pgen_oracle as oracle  # noqa: E402

    shm = Path("/dev/shm") if Path("/dev/shm").is_dir() else Path("/tmp")
    pgen = shm / f"pgenhip_bench_{os.getpid()}.pgen"
    out = shm / f"pgenhip_bench_{os.getpid()}.vcfbody"
    try:
        recs = oracle.synth_records(n_samples, sample_variants, 0, SEED_DATA)
        header = bytes([0x6C, 0x1B, 0x02]) + sample_variants.to_bytes(4, "little") + n_samples.to_bytes(4, "little") + b"\x40"
        with open(pgen, "wb") as f:
            f.write(header)
            f.write(recs.tobytes())
        del recs
        # repeat the bounded sample until ~target_s of CPU work has been timed (output truncated each pass)
        passes, dt = 0, 0.0
        while dt < target_s and passes < 64:
            t0 = time.perf_counter()
            rc = oracle.output_vcf_body_file(str(pgen), n_samples, str(out), n_var=sample_variants)
            dt += time.perf_counter() - t0
            passes += 1
            if rc != 0:
                raise RuntimeError(f"oracle baseline failed: {rc}")
        out_bytes = out.stat().st_size
        assert out_bytes == sample_variants * (4 * n_samples + 1)
    except BaseException:
        for p in (pgen, out):
            try:
                p.unlink()
            except FileNotFoundError:
                pass
        raise
    out.unlink()
    # the same loop on all host cores (variant ranges -> separate files): NOT the reference's behaviour
    # (pgen-rs is single-threaded), reported beside it as BASELINE.md asks
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor

    cores = max(1, min(os.cpu_count() or 1, 64))
    all_cores = None
    try:
        bounds = [sample_variants * i // cores for i in range(cores + 1)]
        idx = [np.arange(bounds[i], bounds[i + 1], dtype=np.uint32) for i in range(cores)]
        outs = [shm / f"pgenhip_bench_{os.getpid()}_{i}.vcfbody" for i in range(cores)]

        def one(i):  # ctypes releases the GIL for the duration of the C call
            return oracle.output_vcf_body_file(str(pgen), n_samples, str(outs[i]), var_idx=idx[i])

        reps, t_all = 0, 0.0
        with ThreadPoolExecutor(max_workers=cores) as ex:
            while t_all < target_s / 3 and reps < 64:
                t0 = time.perf_counter()
                rcs = list(ex.map(one, range(cores)))
                t_all += time.perf_counter() - t0
                reps += 1
                if any(rcs):
                    raise RuntimeError(f"oracle baseline failed: {rcs}")
        all_cores = {"value": reps * sample_variants * n_samples / t_all, "unit": "genotypes/s", "cores": cores,
                     "seconds": t_all, "note": "same C loop, variant ranges on all host cores, one output file per thread; not the reference's behaviour"}
    finally:
        for o in list(locals().get("outs", [])):
            try:
                o.unlink()
            except FileNotFoundError:
                pass
    try:
        pgen.unlink()
    except FileNotFoundError:
        pass
    return {
        "all_cores": all_cores,
        "value": passes * sample_variants * n_samples / dt,
        "unit": "genotypes/s",
        "cores": 1,
        "kind": "port",
        "sample": f"first {sample_variants} variants x {n_samples} samples of the same synthetic workload x {passes} passes, "
                  f".pgen and VCF body on tmpfs, C restatement of src/pfile.rs:149-192 "
                  f"(per-variant alloc+seek+read, two 8-KiB-BufWriter writes per genotype), {dt:.2f} s",
        "seconds": dt,
        "vcf_MB_per_s": passes * out_bytes / dt / 1e6,
        "host_cores_available": os.cpu_count(),
    }


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--variants", type=int, default=CHR22_VARIANTS, help="variants per GPU per step")
    ap.add_argument("--samples", type=int, default=CHR22_SAMPLES)
    ap.add_argument("--keep-modulus", type=int, default=0, help="keep sample i iff splitmix64(seed^i) %% m == 0 (0 = all)")
    ap.add_argument("--cpu-sample-variants", type=int, default=200_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, one rank per GPU) or gloo (rehearsal of the N>1 path)")
    ap.add_argument("--all-ranks-on-device0", action="store_true",
                    help="rehearsal only: every rank uses cuda:0 (with --dist-backend gloo) so the N>1 code path can run on a 1-GPU box")
    ap.add_argument("--kernel", type=int, default=0, help="PGENHIP_KERNEL_* override for A/B runs (0 = automatic, the measured default)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import pgen_rs_amd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
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

    n, v = args.samples, args.variants
    kept = None
    if args.keep_modulus:
        from pgen_rs_amd.synth import keep_indices

        kept = keep_indices(n, modulus=args.keep_modulus)

    eng = pgen_rs_amd.GtEngine(n, kept_idx=kept, device=local_rank)
    k = eng.kept_count
    # every rank owns a distinct block of variants of the same shape (weak scaling, no collective)
    recs = eng.synth_records(v, first_variant=rank * v, seed=SEED_DATA)
    out = torch.empty(v * eng.gt_row_bytes, dtype=torch.uint8, device=dev)

    def step() -> None:
        eng.decode_emit(recs, v, out=out, kernel=args.kernel)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    eng.timer_start()                      # hipEvent on the stream the kernels run on
    for _ in range(args.steps):
        step()
    event_ms = eng.timer_stop()            # records + synchronises the stop event
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0

    t_all = torch.tensor([dt, event_ms], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(t_all, op=dist.ReduceOp.MAX)
    dt_max, event_ms_max = float(t_all[0]), float(t_all[1])

    # spot check on rank 0: the last row is well formed (TAB a / b ... LF); full parity is tests/'s job
    if rank == 0 and v > 0 and k > 0:
        tail = out[(v - 1) * eng.gt_row_bytes :][: eng.gt_row_bytes].cpu()
        assert int(tail[-1]) == 10 and int(tail[0]) == 9 and int(tail[2]) == 47, "malformed GT row"

    if rank == 0:
        genotypes_per_step = world * v * n                    # every 2-bit code of every record is decoded
        alg_bytes_per_launch = v * (eng.record_size + 4 * k + 1)  # R + 4K + 1 per variant
        kernel_ms = event_ms_max / args.steps
        achieved_gbs = alg_bytes_per_launch / (kernel_ms * 1e-3) / 1e9
        traffic = None
        traffic_note = None
        tpath = REPO_ROOT / "profiles" / "r01_hbm_traffic.json"
        if tpath.exists():
            t = json.loads(tpath.read_text())
            if t.get("variants") == v and t.get("samples") == n and t.get("kept") == k:
                traffic = t.get("hbm_bytes_per_launch")
                traffic_note = t.get("note")
        line = {
            "metric": "genotypes decoded/sec (variants x samples / s)",
            "value": genotypes_per_step * args.steps / dt_max,
            "unit": "genotypes/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt_max / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {
                "workload": f"chr22-shape: {v} variants x {n} samples per GPU, {'all samples kept' if kept is None else f'{k} samples kept'}, "
                            "mode-0x02 records resident in HBM -> VCF GT text in HBM (BASELINE.json configs[1])",
                "variants_per_gpu": v,
                "samples": n,
                "kept_samples": k,
                "sharding": f"variant blocks, {world} rank(s), no collective",
            },
            "vcf_MB_per_s": world * v * (4 * k + 1) * args.steps / dt_max / 1e6,
            "genotypes_emitted_per_s": world * v * k * args.steps / dt_max,
            "roofline": {
                "bound": "hbm",
                "achieved": achieved_gbs,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved_gbs / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_note": traffic_note,
                "algorithmic_bytes_per_launch": alg_bytes_per_launch,
                "kernel_ms_avg": kernel_ms,
                "read_only_frac": v * eng.record_size / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(min(args.cpu_sample_variants, v), n)
        print(json.dumps(line), flush=True)

    eng.close()
    if world > 1:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
