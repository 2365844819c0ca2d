"""CPU leg for the multi-GPU path: variant blocks shard with no collective, and the
rank-ordered concatenation of the shards' GT segments is byte-identical to the 1-shard result.
Runs world_size 2 (and 3) over gloo on the CPU; the per-shard decode is the oracle here (no GPU)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import pgen_oracle as oracle
from pgen_rs_amd.sharding import shard_output_offset, shard_range
from pgen_rs_amd.synth import keep_indices


def test_shard_ranges_partition_the_variant_list():
    for n in (0, 1, 7, 8, 9, 1000, 1_000_000):
        for w in (1, 2, 3, 4, 8):
            ranges = [shard_range(n, w, r) for r in range(w)]
            assert ranges[0][0] == 0 and ranges[-1][1] == n
            assert all(ranges[i][1] == ranges[i + 1][0] for i in range(w - 1))
            sizes = [e - b for b, e in ranges]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


def test_keep_indices_matches_oracle_twin():
    for n, m in ((1000, 100), (50001, 3), (500000, 100)):
        assert (keep_indices(n, modulus=m) == oracle.synth_keep(n, modulus=m)).all()


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, n: int, v: int, kept_mod: int, result_path: str):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        kept = keep_indices(n, modulus=kept_mod) if kept_mod else None
        k = n if kept is None else int(kept.size)
        row = 4 * k + 1
        b, e = shard_range(v, world, rank)
        # each rank regenerates only its own block (counter-based generator: no shared file needed)
        recs = oracle.synth_records(n, e - b, first_variant=b)
        part = oracle.decode_emit(recs, e - b, n, kept_idx=kept)
        assert part.size == (e - b) * row
        # timing-style reduction used by bench.py: max over ranks
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert float(t[0]) == float(world)
        # ordered concatenation on rank 0 (host-side, like the CLI): gather by precomputed offsets
        sizes = [(shard_range(v, world, r)[1] - shard_range(v, world, r)[0]) * row for r in range(world)]
        pad = max(sizes)  # gloo gather wants equal sizes: pad the (at most one row) shorter shards
        mine = torch.zeros(pad, dtype=torch.uint8)
        mine[: part.size] = torch.from_numpy(part.copy())
        bufs = [torch.empty(pad, dtype=torch.uint8) for _ in sizes] if rank == 0 else None
        dist.gather(mine, bufs, dst=0)
        if rank == 0:
            whole = np.zeros(v * row, dtype=np.uint8)
            for r, buf in enumerate(bufs):
                off = shard_output_offset(v, world, r, row)
                whole[off : off + sizes[r]] = buf.numpy()[: sizes[r]]
            want = oracle.decode_emit(oracle.synth_records(n, v), v, n, kept_idx=kept)
            np.save(result_path, np.array([int((whole == want).all())]))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n,v,kept_mod", [(2, 2504, 37, 0), (2, 1001, 10, 7), (3, 257, 11, 0)])
def test_sharded_concat_equals_single_shard(tmp_path, world, n, v, kept_mod):
    result = str(tmp_path / "ok.npy")
    mp.spawn(_worker, args=(world, _free_port(), n, v, kept_mod, result), nprocs=world, join=True)
    assert int(np.load(result)[0]) == 1
