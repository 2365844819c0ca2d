"""ctypes wrapper of the CPU oracle (``oracle/libpgen_oracle.so``).

TEST INFRASTRUCTURE ONLY: imported by ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` — never by ``pgen_rs_amd``.  See ``pgen_oracle.h`` for the
reference citations and the "parity unpinned" statement.
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path
from typing import Optional, Sequence

import numpy as np

ORACLE_DIR = Path(__file__).resolve().parent
LIB_PATH = ORACLE_DIR / "libpgen_oracle.so"


def _load() -> C.CDLL:
    if not LIB_PATH.exists():
        subprocess.run(["make", "-C", str(ORACLE_DIR), "libpgen_oracle.so"], check=True, capture_output=True)
    h = C.CDLL(str(LIB_PATH))
    vp = C.c_void_p
    h.pgo_variant_record_size.restype = C.c_uint32
    h.pgo_variant_record_size.argtypes = [C.c_uint32]
    h.pgo_parse_header.restype = C.c_int
    h.pgo_parse_header.argtypes = [C.c_char_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    h.pgo_record_offset_exact.restype = C.c_uint64
    h.pgo_record_offset_exact.argtypes = [C.c_uint64, C.c_uint32]
    h.pgo_record_offset_ref_u32_wrap.restype = C.c_uint64
    h.pgo_record_offset_ref_u32_wrap.argtypes = [C.c_uint64, C.c_uint32]
    h.pgo_decode_emit.restype = C.c_int
    h.pgo_decode_emit.argtypes = [vp, C.c_uint64, vp, C.c_uint32, C.c_uint32, vp, C.c_uint32, vp, C.c_uint64]
    h.pgo_emit_lines.restype = C.c_int
    h.pgo_emit_lines.argtypes = [vp, C.c_uint64, vp, C.c_uint32, C.c_uint32, vp, C.c_uint32, vp, vp, vp, vp]
    h.pgo_output_vcf_body_file.restype = C.c_int
    h.pgo_output_vcf_body_file.argtypes = [C.c_char_p, C.c_uint32, vp, C.c_uint32, vp, C.c_uint32, vp, C.c_char_p, C.c_int, C.c_int]
    h.pgo_splitmix64.restype = C.c_uint64
    h.pgo_splitmix64.argtypes = [C.c_uint64]
    h.pgo_synth_records.restype = None
    h.pgo_synth_records.argtypes = [vp, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint64, C.c_int]
    h.pgo_synth_records_hwe.restype = None
    h.pgo_synth_records_hwe.argtypes = [vp, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint64]
    h.pgo_synth_keep.restype = C.c_uint32
    h.pgo_synth_keep.argtypes = [C.c_uint32, C.c_uint64, C.c_uint32, vp, C.c_uint32]
    # variable-width header walk (pgen_vw_oracle.c)
    h.pgo_vw_parse_header.restype = C.c_int
    h.pgo_vw_parse_header.argtypes = [C.c_char_p, C.POINTER(VwHeader)]
    for name in ("pgo_vw_variant_block_count", "pgo_vw_main_header_body_offset", "pgo_vw_main_header_body_size", "pgo_vw_variant_records_offset"):
        getattr(h, name).restype = C.c_uint64
        getattr(h, name).argtypes = [C.POINTER(VwHeader)]
    h.pgo_vw_check_variant_block_offsets.restype = C.c_int64
    h.pgo_vw_check_variant_block_offsets.argtypes = [C.POINTER(VwHeader), vp, C.c_uint64, C.c_uint64]
    h.pgo_vw_check_main_header_body.restype = C.c_int64
    h.pgo_vw_check_main_header_body.argtypes = [C.POINTER(VwHeader), vp, C.c_uint64, C.c_uint64, vp, vp]
    h.pgo_vw_index.restype = C.c_int
    h.pgo_vw_index.argtypes = [C.POINTER(VwHeader), vp, C.c_uint64, vp, vp, vp]
    h.pgo_decode_emit_at.restype = C.c_int
    h.pgo_decode_emit_at.argtypes = [vp, vp, C.c_uint32, C.c_uint32, vp, C.c_uint32, vp, C.c_uint64]
    return h


class VwHeader(C.Structure):
    """``pgo_vw_header``: what src/pgen.rs:21-98 parses out of the 12 header bytes."""
    _fields_ = [("storage_mode", C.c_uint8), ("variant_count", C.c_uint32), ("sample_count", C.c_uint32),
                ("record_type_bits", C.c_uint8), ("record_length_bytes", C.c_uint8), ("allele_count_bytes", C.c_uint8),
                ("provisional_ref_storage", C.c_uint8)]


lib = _load()


def _vp(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _u32(a) -> Optional[np.ndarray]:
    return None if a is None else np.ascontiguousarray(np.asarray(a, dtype=np.uint32))


def variant_record_size(n: int) -> int:
    return int(lib.pgo_variant_record_size(n))


def parse_header(hdr: bytes):
    nv, ns = C.c_uint32(), C.c_uint32()
    rc = lib.pgo_parse_header(bytes(hdr[:12]), C.byref(nv), C.byref(ns))
    return rc, nv.value, ns.value


def record_offset_exact(v: int, r: int) -> int:
    return int(lib.pgo_record_offset_exact(v, r))


def record_offset_ref_u32_wrap(v: int, r: int) -> int:
    return int(lib.pgo_record_offset_ref_u32_wrap(v, r))


def decode_emit(records: np.ndarray, n_variants: int, num_samples: int, kept_idx: Optional[Sequence[int]] = None,
                record_stride: Optional[int] = None, variant_idx: Optional[Sequence[int]] = None,
                out_stride: Optional[int] = None, records_offset: int = 0) -> np.ndarray:
    """Oracle GT segments for a block; returns uint8 array of n_variants*out_stride bytes (gaps zero)."""
    records = np.ascontiguousarray(records, dtype=np.uint8)
    R = variant_record_size(num_samples)
    if record_stride is None:
        record_stride = R
    kept = _u32(kept_idx)
    K = num_samples if kept is None else int(kept.size)
    if out_stride is None:
        out_stride = 4 * K + 1
    vidx = _u32(variant_idx)
    out = np.zeros(max(n_variants * out_stride, 1), dtype=np.uint8)
    base = records[records_offset:]
    rc = lib.pgo_decode_emit(_vp(base), record_stride, _vp(vidx), n_variants, num_samples, _vp(kept), K, _vp(out), out_stride)
    if rc != 0:
        raise IndexError(f"oracle decode_emit failed: {rc}")
    return out[: n_variants * out_stride]


def emit_lines(records: np.ndarray, n_variants: int, num_samples: int, prefix_blob: np.ndarray, prefix_off: np.ndarray,
               line_off: np.ndarray, kept_idx=None, record_stride=None, variant_idx=None) -> np.ndarray:
    records = np.ascontiguousarray(records, dtype=np.uint8)
    R = variant_record_size(num_samples)
    if record_stride is None:
        record_stride = R
    kept = _u32(kept_idx)
    K = num_samples if kept is None else int(kept.size)
    vidx = _u32(variant_idx)
    prefix_blob = np.ascontiguousarray(prefix_blob, dtype=np.uint8)
    prefix_off = np.ascontiguousarray(prefix_off, dtype=np.uint64)
    line_off = np.ascontiguousarray(line_off, dtype=np.uint64)
    out = np.zeros(max(int(line_off[n_variants]), 1), dtype=np.uint8)
    rc = lib.pgo_emit_lines(_vp(records), record_stride, _vp(vidx), n_variants, num_samples, _vp(kept), K,
                            _vp(prefix_blob), _vp(prefix_off), _vp(line_off), _vp(out))
    if rc != 0:
        raise ValueError(f"oracle emit_lines failed: {rc}")
    return out[: int(line_off[n_variants])]


def output_vcf_body_file(pgen_path: str, num_samples: int, out_path: str, var_idx=None, n_var: Optional[int] = None,
                         kept_idx=None, prefixes: Optional[Sequence[bytes]] = None, append: bool = False,
                         wrap_u32: bool = False) -> int:
    v = _u32(var_idx)
    if n_var is None:
        n_var = int(v.size)
    kept = _u32(kept_idx)
    K = num_samples if kept is None else int(kept.size)
    parr = None
    if prefixes is not None:
        parr = (C.c_char_p * len(prefixes))(*prefixes)
    return int(lib.pgo_output_vcf_body_file(pgen_path.encode(), num_samples, _vp(v), n_var, _vp(kept), K,
                                            C.cast(parr, C.c_void_p) if parr is not None else None,
                                            out_path.encode(), int(append), int(wrap_u32)))


def splitmix64(x: int) -> int:
    return int(lib.pgo_splitmix64(x & 0xFFFFFFFFFFFFFFFF))


def synth_records(num_samples: int, n_variants: int, first_variant: int = 0, seed: int = 0x5047454E,
                  record_stride: Optional[int] = None, dirty_pad: bool = False, hwe: bool = False) -> np.ndarray:
    R = variant_record_size(num_samples)
    if record_stride is None:
        record_stride = R
    dst = np.zeros(max(n_variants * record_stride, 1), dtype=np.uint8)
    if hwe:
        lib.pgo_synth_records_hwe(_vp(dst), record_stride, num_samples, first_variant, n_variants, seed)
    else:
        lib.pgo_synth_records(_vp(dst), record_stride, num_samples, first_variant, n_variants, seed, int(dirty_pad))
    return dst[: n_variants * record_stride]


def synth_keep(num_samples: int, seed: int = 0x4D41534B, modulus: int = 100) -> np.ndarray:
    buf = np.zeros(max(num_samples, 1), dtype=np.uint32)
    n = lib.pgo_synth_keep(num_samples, seed, modulus, _vp(buf), num_samples)
    return buf[:n].copy()


# ---- variable-width storage modes (src/pgen.rs) -------------------------------------------------------------
def vw_parse_header(hdr: bytes):
    """-> (rc, VwHeader); rc as pgo_vw_parse_header (0 ok, -1 magic, -2 provisional ref, -3 record storage mode)."""
    h = VwHeader()
    rc = lib.pgo_vw_parse_header(bytes(hdr[:12]), C.byref(h))
    return int(rc), h


def vw_geometry(h: VwHeader) -> dict:
    return {
        "block_count": int(lib.pgo_vw_variant_block_count(C.byref(h))),
        "main_header_body_offset": int(lib.pgo_vw_main_header_body_offset(C.byref(h))),
        "main_header_body_size": int(lib.pgo_vw_main_header_body_size(C.byref(h))),
        "variant_records_offset": int(lib.pgo_vw_variant_records_offset(C.byref(h))),
    }


def vw_validate(h: VwHeader, file_bytes: bytes):
    """The reference's two walks (src/pgen.rs:140-258) -> (pos after block offsets, pos after header body, types seen, length bytes seen)."""
    buf = np.frombuffer(file_bytes, dtype=np.uint8)
    p1 = int(lib.pgo_vw_check_variant_block_offsets(C.byref(h), _vp(buf), buf.size, 12))
    types = np.zeros(256, dtype=np.uint8)
    lens = np.zeros(256, dtype=np.uint8)
    p2 = int(lib.pgo_vw_check_main_header_body(C.byref(h), _vp(buf), buf.size, max(p1, 0), _vp(types), _vp(lens)))
    return p1, p2, sorted(np.nonzero(types)[0].tolist()), sorted(np.nonzero(lens)[0].tolist())


def vw_index(h: VwHeader, file_bytes: bytes):
    """-> (rc, types u8[V], lens u32[V], offs u64[V]); not in the reference (parity unpinned)."""
    buf = np.frombuffer(file_bytes, dtype=np.uint8)
    v = int(h.variant_count)
    types = np.zeros(max(v, 1), dtype=np.uint8)
    lens = np.zeros(max(v, 1), dtype=np.uint32)
    offs = np.zeros(max(v, 1), dtype=np.uint64)
    rc = int(lib.pgo_vw_index(C.byref(h), _vp(buf), buf.size, _vp(types), _vp(lens), _vp(offs)))
    return rc, types[:v], lens[:v], offs[:v]


def decode_emit_at(base: np.ndarray, record_off, num_samples: int, kept_idx=None) -> np.ndarray:
    """GT segments of the records at byte offsets ``record_off`` of ``base`` (dense 4K+1 pitch)."""
    base = np.ascontiguousarray(base, dtype=np.uint8)
    off = np.ascontiguousarray(np.asarray(record_off, dtype=np.uint64))
    kept = _u32(kept_idx)
    K = num_samples if kept is None else int(kept.size)
    out = np.zeros(max(off.size * (4 * K + 1), 1), dtype=np.uint8)
    rc = lib.pgo_decode_emit_at(_vp(base), _vp(off), off.size, num_samples, _vp(kept), K, _vp(out), 4 * K + 1)
    if rc != 0:
        raise IndexError(f"oracle decode_emit_at failed: {rc}")
    return out[: off.size * (4 * K + 1)]
