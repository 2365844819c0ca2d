"""pgen_rs_amd — MI355X (gfx950) engine for the GT decode/emit hot path of teoremma/pgen-rs.

Layout: ``csrc/`` (hand-written HIP kernels + the C ABI of ``include/pgen_hip.h``), ``host/``
(C++ restatement of the reference's Pfile/CLI surface above that ABI) and thin Python plumbing
(``engine``) used by the tests and ``bench.py``.  Importing the package requires the built
``libpgen_hip.so``; there is no CPU fallback.
"""
from ._capi import (  # noqa: F401
    KERNEL_AUTO,
    KERNEL_FLAT,
    KERNEL_ROWS,
    KERNEL_SCAN,
    KERNEL_PICK,
    KERNEL_RUNS,
    KERNEL_WIDE,
    PgenHipError,
)
from .engine import (  # noqa: F401
    GtEngine,
    device_count,
    parse_header,
    record_offset,
    variant_record_size,
    vw_parse_header,
    vw_select_uncompressed,
    vw_walk_index,
)

__all__ = [
    "GtEngine",
    "PgenHipError",
    "device_count",
    "parse_header",
    "record_offset",
    "variant_record_size",
    "vw_parse_header",
    "vw_walk_index",
    "vw_select_uncompressed",
    "KERNEL_AUTO",
    "KERNEL_ROWS",
    "KERNEL_FLAT",
    "KERNEL_SCAN",
    "KERNEL_WIDE",
    "KERNEL_PICK",
    "KERNEL_RUNS",
]
