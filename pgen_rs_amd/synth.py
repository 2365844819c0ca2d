"""Synthetic-workload helpers on the host side: the keep-mask generator used by
bench.py and the CLI's synthetic fixtures.  Record bytes themselves are generated on the device
(``GtEngine.synth_records``)."""
from __future__ import annotations

import numpy as np

SEED_DATA = 0x5047454E  # "PGEN"
SEED_MASK = 0x4D41534B  # "MASK"
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x: np.ndarray) -> np.ndarray:
    """Vectorised SplitMix64 finaliser on uint64 arrays."""
    with np.errstate(over="ignore"):
        z = (x.astype(np.uint64) + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def keep_indices(sample_count: int, seed: int = SEED_MASK, modulus: int = 100) -> np.ndarray:
    """Ascending indices i with splitmix64(seed ^ i) % modulus == 0 (config 5: modulus 100 ~ 1 % kept)."""
    i = np.arange(sample_count, dtype=np.uint64)
    return np.nonzero(splitmix64(np.uint64(seed) ^ i) % np.uint64(modulus) == 0)[0].astype(np.uint32)
