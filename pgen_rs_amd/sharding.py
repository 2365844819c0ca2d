"""Variant-block sharding across GPUs: contiguous ranges of the kept-variant
list, one per rank, no data-path collective — every GT segment is 4K+1 bytes, so rank r's output
lands at a precomputed offset and the host (or rank 0) concatenates in rank order.

Reference: the outer loop of ``Pfile::output_vcf`` (/root/reference/src/pfile.rs:156) iterates
the kept variants in file order; a shard is a contiguous slice of that iteration space.

There is ONE partitioner, ``pgenhip_shard_range`` in the C ABI: the C++ host's per-device threads
(``host/pfile.cpp``), ``bench.py``'s ranks and the tests all go through it.
"""
from __future__ import annotations

import ctypes as C

from ._capi import lib


def shard_range(n_variants: int, world_size: int, rank: int) -> tuple[int, int]:
    """[begin, end) of the kept-variant list owned by ``rank``; sizes differ by at most one."""
    if world_size <= 0 or not 0 <= rank < world_size or n_variants < 0:
        raise ValueError(f"bad rank {rank} / world {world_size}")
    b, e = C.c_uint64(), C.c_uint64()
    if lib.pgenhip_shard_range(n_variants, world_size, rank, C.byref(b), C.byref(e)) != 0:
        raise ValueError(f"bad rank {rank} / world {world_size}")
    return int(b.value), int(e.value)


def shard_output_offset(n_variants: int, world_size: int, rank: int, gt_row_bytes: int) -> int:
    """Byte offset of rank's first GT segment in the concatenated body (dense 4K+1 pitch)."""
    begin, _ = shard_range(n_variants, world_size, rank)
    return begin * gt_row_bytes
