"""Variant-block sharding across GPUs: contiguous ranges of the kept-variant
list, one per rank, no data-path collective — every GT segment is 4K+1 bytes, so rank r's output
lands at a precomputed offset and the host (or rank 0) concatenates in rank order.

Reference: the outer loop of ``Pfile::output_vcf`` (/root/reference/src/pfile.rs:156) iterates
the kept variants in file order; a shard is a contiguous slice of that iteration space.
"""
from __future__ import annotations


def shard_range(n_variants: int, world_size: int, rank: int) -> tuple[int, int]:
    """[begin, end) of the kept-variant list owned by ``rank``; sizes differ by at most one."""
    if world_size <= 0 or not 0 <= rank < world_size:
        raise ValueError(f"bad rank {rank} / world {world_size}")
    base, extra = divmod(n_variants, world_size)
    begin = rank * base + min(rank, extra)
    end = begin + base + (1 if rank < extra else 0)
    return begin, end


def shard_output_offset(n_variants: int, world_size: int, rank: int, gt_row_bytes: int) -> int:
    """Byte offset of rank's first GT segment in the concatenated body (dense 4K+1 pitch)."""
    begin, _ = shard_range(n_variants, world_size, rank)
    return begin * gt_row_bytes
