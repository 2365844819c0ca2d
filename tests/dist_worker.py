"""Worker of tests/test_sharding_gpu.py: one rank of a `python -m torch.distributed.run` job.

Each rank takes its contiguous slice of the kept-variant list from the ONE partitioner
(pgen_rs_amd.sharding.shard_range -> pgenhip_shard_range), decodes it with the HIP engine (GtEngine, through
the C ABI) on the device it is given and writes its GT segments to `part<rank>.bin`; nothing is exchanged
between ranks on the data path — the process group only carries a barrier and the
max-over-ranks reduction bench.py uses.  No oracle in here: the test process does the comparison.
"""
import argparse
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--out-dir", required=True)
    ap.add_argument("--samples", type=int, required=True)
    ap.add_argument("--variants", type=int, required=True)
    ap.add_argument("--keep-modulus", type=int, default=0)
    ap.add_argument("--all-ranks-on-device0", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import pgen_rs_amd
    from pgen_rs_amd.sharding import shard_output_offset, shard_range
    from pgen_rs_amd.synth import keep_indices

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = 0 if args.all_ranks_on_device0 else int(os.environ.get("LOCAL_RANK", "0"))
    dist.init_process_group("gloo")
    torch.cuda.set_device(local)
    kept = keep_indices(args.samples, modulus=args.keep_modulus) if args.keep_modulus else None
    b, e = shard_range(args.variants, world, rank)
    with pgen_rs_amd.GtEngine(args.samples, kept_idx=kept, device=local) as eng:
        recs = eng.synth_records(e - b, first_variant=b)
        out = eng.decode_emit(recs, e - b)
        eng.wait()
        part = out[: (e - b) * eng.gt_row_bytes].cpu().numpy()
        off = shard_output_offset(args.variants, world, rank, eng.gt_row_bytes)
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert float(t[0]) == float(world)
    Path(args.out_dir, f"part{rank}.bin").write_bytes(part.tobytes())
    Path(args.out_dir, f"part{rank}.off").write_text(str(off))
    dist.barrier()
    dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
