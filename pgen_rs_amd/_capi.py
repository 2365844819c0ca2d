"""ctypes binding of ``libpgen_hip.so`` (the C ABI in ``include/pgen_hip.h``).

There is no fallback: if the shared library is missing this module raises at import, and if
no HIP device is usable ``pgenhip_create`` returns ``PGENHIP_ERR_NO_DEVICE``.

torch is imported first on purpose: the PyTorch-ROCm wheel bundles its own
``libamdhip64.so.7``; loading it before our library makes the dynamic linker bind our
``DT_NEEDED libamdhip64.so.7`` to that same runtime, so device pointers from torch tensors and
our kernel launches live in one HIP runtime instance.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import torch  # noqa: F401  (must precede CDLL — see module docstring)

PKG_DIR = Path(__file__).resolve().parent
LIB_PATH = PKG_DIR / "libpgen_hip.so"

OK = 0
ERR_BAD_ARG = -1
ERR_HIP = -2
ERR_OOM = -3
ERR_INDEX_RANGE = -4
ERR_BAD_MAGIC = -5
ERR_BAD_MODE = -6
ERR_BAD_FLAGS = -7
ERR_NO_DEVICE = -8
ERR_TOO_LARGE = -9
ERR_IO = -10
ERR_BAD_INDEX = -11
ERR_COMPRESSED_RECORD = -12

KERNEL_AUTO = 0
KERNEL_ROWS = 1
KERNEL_FLAT = 2
KERNEL_SCAN = 3
KERNEL_WIDE = 4
KERNEL_PICK = 6
KERNEL_RUNS = 7
KERNEL_ROWPICK = 8
SYNTH_DIRTY_PAD = 1
SYNTH_HWE = 2
CREATE_KEEP_LIST = 1
LAUNCHES_IN_FLIGHT = 16

KNOB_WIDE_BLOCKS_PER_CU = 1
KNOB_WIDE_RANGES = 2
KNOB_FLAT_BLOCKS_PER_CU = 3
KNOB_SCAN_BLOCKS_PER_CU = 4
KNOB_PICK_BATCH_BYTES = 6
KNOB_RUNS_ROWS = 7
KNOB_SCAN_XCD_MAP = 8
KNOB_SCAN_TWO_PASS = 9
KNOB_SCAN_CHUNK_ROWS = 10
KNOB_ROWPICK_BLOCKS_PER_CU = 11
KNOB_SCAN_ROWPICK = 12
KNOB_PICK_LINE_SEAMS = 13
KNOB_FLUSH_UNROLL = 14
KNOB_SCAN_FOUR_PICKS = 15
KNOB_ALIGN_STORES = 16



class VwHeader(C.Structure):
    """``pgenhip_vw_header`` (include/pgen_hip.h): header of a variable-width .pgen (src/pgen.rs:21-137)."""
    _fields_ = [("variant_count", C.c_uint32), ("sample_count", C.c_uint32), ("storage_mode", C.c_uint8),
                ("record_type_bits", C.c_uint8), ("record_length_bytes", C.c_uint8), ("allele_count_bytes", C.c_uint8),
                ("provisional_ref_storage", C.c_uint8), ("reserved", C.c_uint8 * 3), ("block_count", C.c_uint64),
                ("main_header_body_offset", C.c_uint64), ("variant_records_offset", C.c_uint64)]


u8p = C.POINTER(C.c_uint8)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)
ctx_p = C.c_void_p

# name -> (restype, argtypes); every symbol include/pgen_hip.h declares
PROTOTYPES = {
    "pgenhip_abi_version": (C.c_uint32, []),
    "pgenhip_strerror": (C.c_char_p, [C.c_int]),
    "pgenhip_last_error_detail": (C.c_char_p, []),
    "pgenhip_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "pgenhip_variant_record_size": (C.c_uint32, [C.c_uint32]),
    "pgenhip_parse_header": (C.c_int, [C.c_char_p, u32p, u32p]),
    "pgenhip_record_offset": (C.c_uint64, [C.c_uint64, C.c_uint32]),
    "pgenhip_shard_range": (C.c_int, [C.c_uint64, C.c_uint32, C.c_uint32, u64p, u64p]),
    "pgenhip_vw_parse_header": (C.c_int, [C.c_char_p, C.POINTER(VwHeader)]),
    "pgenhip_vw_walk_index": (C.c_int, [C.POINTER(VwHeader), C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pgenhip_vw_select_uncompressed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]),
    "pgenhip_create": (C.c_int, [C.POINTER(ctx_p), C.c_int, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32]),
    "pgenhip_destroy": (C.c_int, [ctx_p]),
    "pgenhip_set_stream": (C.c_int, [ctx_p, C.c_void_p]),
    "pgenhip_reset_stream": (C.c_int, [ctx_p]),
    "pgenhip_sample_count": (C.c_uint32, [ctx_p]),
    "pgenhip_kept_count": (C.c_uint32, [ctx_p]),
    "pgenhip_gt_row_bytes": (C.c_uint64, [ctx_p]),
    "pgenhip_decode_emit": (C.c_int, [ctx_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64, C.c_uint32]),
    "pgenhip_decode_emit_at": (C.c_int, [ctx_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64, C.c_uint32]),
    "pgenhip_emit_lines": (C.c_int, [ctx_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32]),
    "pgenhip_tune": (C.c_int, [ctx_p, C.c_uint32, C.c_int32]),
    "pgenhip_wait": (C.c_int, [ctx_p]),
    "pgenhip_timer_start": (C.c_int, [ctx_p]),
    "pgenhip_timer_stop": (C.c_int, [ctx_p, C.POINTER(C.c_float)]),
    "pgenhip_timer_mark": (C.c_int, [ctx_p]),
    "pgenhip_timer_read": (C.c_int, [ctx_p, C.POINTER(C.c_float)]),
    "pgenhip_device_malloc": (C.c_int, [ctx_p, C.POINTER(C.c_void_p), C.c_size_t]),
    "pgenhip_device_free": (C.c_int, [ctx_p, C.c_void_p]),
    "pgenhip_host_malloc_pinned": (C.c_int, [ctx_p, C.POINTER(C.c_void_p), C.c_size_t]),
    "pgenhip_host_free_pinned": (C.c_int, [ctx_p, C.c_void_p]),
    "pgenhip_memcpy_h2d": (C.c_int, [ctx_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "pgenhip_memcpy_d2h": (C.c_int, [ctx_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "pgenhip_synth_records": (C.c_int, [ctx_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint32]),
}


class PgenHipError(RuntimeError):
    def __init__(self, status: int, where: str):
        self.status = status
        detail = lib.pgenhip_last_error_detail().decode(errors="replace")
        msg = lib.pgenhip_strerror(status).decode()
        super().__init__(f"{where}: {msg} (status {status}){': ' + detail if detail else ''}")


def _load() -> C.CDLL:
    if not LIB_PATH.exists():
        raise RuntimeError(
            f"{LIB_PATH} is missing — the HIP extension is not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). There is no CPU fallback."
        )
    handle = C.CDLL(str(LIB_PATH), mode=C.RTLD_GLOBAL)
    for name, (restype, argtypes) in PROTOTYPES.items():
        fn = getattr(handle, name)  # AttributeError if the symbol is not exported
        fn.restype = restype
        fn.argtypes = argtypes
    return handle


lib = _load()


def check(status: int, where: str) -> None:
    if status != OK:
        raise PgenHipError(status, where)
