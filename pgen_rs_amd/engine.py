"""Python face of the GT decode/emit engine: a thin wrapper over the C ABI (``include/pgen_hip.h``).

PyTorch is plumbing only (device buffers, streams, process groups); every byte of output is
produced by the gfx950 kernels inside ``libpgen_hip.so``.  Nothing here computes genotypes on
the CPU and nothing here touches ``oracle/``.

Reference seam: ``/root/reference/src/pfile.rs:156-192`` (``Pfile::output_vcf`` hot loop).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np
import torch

from . import _capi
from ._capi import KERNEL_AUTO, check, lib


def variant_record_size(sample_count: int) -> int:
    """``Pfile::variant_record_size`` (src/pfile.rs:196-200) through the C ABI."""
    return int(lib.pgenhip_variant_record_size(sample_count))


def record_offset(var_idx: int, record_size: int) -> int:
    """Byte offset of record ``var_idx`` (src/pfile.rs:165, u64 — see SURVEY.md F5)."""
    return int(lib.pgenhip_record_offset(var_idx, record_size))


def parse_header(header: bytes) -> tuple[int, int]:
    """12-byte .pgen header -> (variant_count, sample_count); raises on the reference's asserts (src/pfile.rs:47,53,69)."""
    if len(header) < 12:
        raise _capi.PgenHipError(_capi.ERR_IO, "parse_header: short header")
    nv, ns = C.c_uint32(), C.c_uint32()
    check(lib.pgenhip_parse_header(bytes(header[:12]), C.byref(nv), C.byref(ns)), "pgenhip_parse_header")
    return nv.value, ns.value


def device_count() -> int:
    n = C.c_int()
    rc = lib.pgenhip_device_count(C.byref(n))
    return n.value if rc == 0 else 0


# ---- variable-width storage modes: header and offset-table walk (src/pgen.rs) ---------
def vw_parse_header(header: bytes) -> "_capi.VwHeader":
    """12 header bytes of a variable-width .pgen -> ``pgenhip_vw_header`` (src/pgen.rs:21-137); raises on its asserts."""
    if len(header) < 12:
        raise _capi.PgenHipError(_capi.ERR_IO, "vw_parse_header: short header")
    h = _capi.VwHeader()
    check(lib.pgenhip_vw_parse_header(bytes(header[:12]), C.byref(h)), "pgenhip_vw_parse_header")
    return h


def vw_walk_index(h: "_capi.VwHeader", index: bytes):
    """File bytes [12, variant_records_offset) -> (record_type u8[V], record_len u32[V], record_off u64[V]) (src/pgen.rs:140-258)."""
    v = int(h.variant_count)
    types = np.zeros(max(v, 1), dtype=np.uint8)
    lens = np.zeros(max(v, 1), dtype=np.uint32)
    offs = np.zeros(max(v, 1), dtype=np.uint64)
    buf = np.frombuffer(bytes(index), dtype=np.uint8)
    check(lib.pgenhip_vw_walk_index(C.byref(h), buf.ctypes.data_as(C.c_void_p) if buf.size else None, buf.size,
                                    types.ctypes.data_as(C.c_void_p), lens.ctypes.data_as(C.c_void_p), offs.ctypes.data_as(C.c_void_p)),
          "pgenhip_vw_walk_index")
    return types[:v], lens[:v], offs[:v]


def vw_select_uncompressed(types: np.ndarray, lens: np.ndarray, offs: np.ndarray, record_size: int,
                           variant_idx: Optional[Sequence[int]] = None) -> np.ndarray:
    """Byte offsets of the selected variants' records, all of which must be plain 2-bit records (type 0, length R);
    raises ``PgenHipError`` with status ``ERR_COMPRESSED_RECORD`` otherwise."""
    types = np.ascontiguousarray(types, dtype=np.uint8)
    lens = np.ascontiguousarray(lens, dtype=np.uint32)
    offs = np.ascontiguousarray(offs, dtype=np.uint64)
    vidx = None if variant_idx is None else np.ascontiguousarray(np.asarray(variant_idx, dtype=np.uint32))
    n = int(types.size if vidx is None else vidx.size)
    sel = np.zeros(max(n, 1), dtype=np.uint64)
    check(lib.pgenhip_vw_select_uncompressed(types.ctypes.data_as(C.c_void_p), lens.ctypes.data_as(C.c_void_p), offs.ctypes.data_as(C.c_void_p),
                                             int(types.size), vidx.ctypes.data_as(C.c_void_p) if vidx is not None and vidx.size else None, n,
                                             record_size, sel.ctypes.data_as(C.c_void_p)), "pgenhip_vw_select_uncompressed")
    return sel[:n]


def _ptr(t: Optional[torch.Tensor], byte_offset: int = 0) -> Optional[int]:
    if t is None:
        return None
    return t.data_ptr() + byte_offset


class GtEngine:
    """One context = one device + one kept-sample list (src/pfile.rs:128 ``sam_idx_rcs``).

    ``kept_idx=None`` keeps all samples (K = N).  Launches are asynchronous on torch's current
    stream of the device (``use_torch_stream``), so torch ops and kernels are stream-ordered.
    """

    def __init__(self, sample_count: int, kept_idx: Optional[Sequence[int]] = None, device: int = 0):
        self._ctx = C.c_void_p()
        self.device = int(device)
        self.sample_count = int(sample_count)
        arr = None
        if kept_idx is not None:
            arr = np.ascontiguousarray(np.asarray(kept_idx, dtype=np.uint32))
        check(
            lib.pgenhip_create(
                C.byref(self._ctx),
                self.device,
                self.sample_count,
                arr.ctypes.data_as(C.c_void_p) if arr is not None and arr.size else None,
                int(arr.size) if arr is not None else 0,
                _capi.CREATE_KEEP_LIST if arr is not None else 0,  # an empty list is a list, not "all samples"
            ),
            "pgenhip_create",
        )
        self.kept_count = int(lib.pgenhip_kept_count(self._ctx))
        self.record_size = variant_record_size(self.sample_count)
        self.gt_row_bytes = int(lib.pgenhip_gt_row_bytes(self._ctx))
        self.torch_device = torch.device("cuda", self.device)
        self.use_torch_stream()

    # -- lifetime ---------------------------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "_ctx", None) is not None and self._ctx.value:
            lib.pgenhip_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- streams / timing -------------------------------------------------------------------
    def use_torch_stream(self) -> None:
        stream = torch.cuda.current_stream(self.torch_device)
        check(lib.pgenhip_set_stream(self._ctx, C.c_void_p(stream.cuda_stream)), "pgenhip_set_stream")

    def use_own_stream(self) -> None:
        check(lib.pgenhip_reset_stream(self._ctx), "pgenhip_reset_stream")

    def use_stream(self, stream: "torch.cuda.Stream") -> None:
        check(lib.pgenhip_set_stream(self._ctx, C.c_void_p(stream.cuda_stream)), "pgenhip_set_stream")

    def tune(self, knob: int, value: int) -> None:
        """Launch-shape knob of this context (``_capi.KNOB_*``): tests force small grids, probes A/B."""
        check(lib.pgenhip_tune(self._ctx, knob, value), "pgenhip_tune")

    def wait(self) -> None:
        check(lib.pgenhip_wait(self._ctx), "pgenhip_wait")

    def timer_start(self) -> None:
        check(lib.pgenhip_timer_start(self._ctx), "pgenhip_timer_start")

    def timer_stop(self) -> float:
        ms = C.c_float()
        check(lib.pgenhip_timer_stop(self._ctx, C.byref(ms)), "pgenhip_timer_stop")
        return float(ms.value)

    # -- the hot path -----------------------------------------------------------------------
    def decode_emit(
        self,
        records: torch.Tensor,
        n_variants: int,
        record_stride: Optional[int] = None,
        variant_idx: Optional[torch.Tensor] = None,
        out: Optional[torch.Tensor] = None,
        out_stride: Optional[int] = None,
        kernel: int = KERNEL_AUTO,
        records_offset: int = 0,
        out_offset: int = 0,
    ) -> torch.Tensor:
        """GT segments of ``n_variants`` rows (src/pfile.rs:165-190); returns the uint8 output tensor.

        ``records``: uint8 CUDA tensor; row r at byte ``records_offset + r*record_stride``.
        ``variant_idx``: optional int32/uint32 CUDA tensor of row numbers (gapped kept-variant lists).
        Row j is written at byte ``out_offset + j*out_stride`` of ``out`` (dense 4K+1 by default).
        """
        if record_stride is None:
            record_stride = self.record_size
        if out_stride is None:
            out_stride = self.gt_row_bytes
        if out is None:
            out = torch.empty(out_offset + n_variants * out_stride, dtype=torch.uint8, device=self.torch_device)
        self._check_dev(records, "records")
        self._check_dev(out, "out")
        if n_variants:
            need_out = out_offset + (n_variants - 1) * out_stride + self.gt_row_bytes
            if out.numel() < need_out:
                raise ValueError(f"out too small: {out.numel()} < {need_out}")
            if variant_idx is None:
                need_in = records_offset + (n_variants - 1) * record_stride + self.record_size
                if records.numel() < need_in:
                    raise ValueError(f"records too small: {records.numel()} < {need_in}")
        if variant_idx is not None:
            self._check_dev(variant_idx, "variant_idx")
            if variant_idx.dtype not in (torch.int32, torch.uint32) or variant_idx.numel() < n_variants:
                raise ValueError("variant_idx must be a 32-bit integer tensor with >= n_variants entries")
        check(
            lib.pgenhip_decode_emit(
                self._ctx,
                _ptr(records, records_offset),
                record_stride,
                _ptr(variant_idx),
                n_variants,
                _ptr(out, out_offset),
                out_stride,
                kernel,
            ),
            "pgenhip_decode_emit",
        )
        return out

    def decode_emit_at(self, base: torch.Tensor, record_off: torch.Tensor, n_variants: int, out: Optional[torch.Tensor] = None,
                       out_stride: Optional[int] = None, kernel: int = KERNEL_AUTO) -> torch.Tensor:
        """GT segments of records addressed by BYTE OFFSET into ``base`` (``record_off``: int64 CUDA tensor): the
        uncompressed records of a variable-width .pgen staged to HBM as it lies on disk."""
        if out_stride is None:
            out_stride = self.gt_row_bytes
        if out is None:
            out = torch.empty(max(n_variants, 1) * out_stride, dtype=torch.uint8, device=self.torch_device)
        for t, name in ((base, "base"), (record_off, "record_off"), (out, "out")):
            self._check_dev(t, name)
        if record_off.dtype != torch.int64 or record_off.numel() < n_variants:
            raise ValueError("record_off must be an int64 tensor (u64 byte offsets) with >= n_variants entries")
        if n_variants and out.numel() < (n_variants - 1) * out_stride + self.gt_row_bytes:
            raise ValueError("out too small")
        check(lib.pgenhip_decode_emit_at(self._ctx, _ptr(base), _ptr(record_off), n_variants, _ptr(out), out_stride, kernel), "pgenhip_decode_emit_at")
        return out

    def emit_lines(
        self,
        records: torch.Tensor,
        n_variants: int,
        prefix_blob: torch.Tensor,
        prefix_off: torch.Tensor,
        line_off: torch.Tensor,
        max_prefix_bytes: int,
        out: torch.Tensor,
        record_stride: Optional[int] = None,
        variant_idx: Optional[torch.Tensor] = None,
        kernel: int = KERNEL_AUTO,
        records_offset: int = 0,
    ) -> torch.Tensor:
        """Complete VCF body lines (src/pfile.rs:156-192): prefix + GT segment + newline per variant."""
        if record_stride is None:
            record_stride = self.record_size
        for t, name in ((records, "records"), (prefix_blob, "prefix_blob"), (prefix_off, "prefix_off"), (line_off, "line_off"), (out, "out")):
            self._check_dev(t, name)
        if prefix_off.dtype != torch.int64 or line_off.dtype != torch.int64:
            raise ValueError("prefix_off/line_off must be int64 tensors (u64 offsets)")
        if prefix_off.numel() < n_variants + 1 or line_off.numel() < n_variants + 1:
            raise ValueError("offset arrays need n_variants+1 entries")
        check(
            lib.pgenhip_emit_lines(
                self._ctx,
                _ptr(records, records_offset),
                record_stride,
                _ptr(variant_idx),
                n_variants,
                _ptr(prefix_blob),
                _ptr(prefix_off),
                _ptr(line_off),
                max_prefix_bytes,
                _ptr(out),
                kernel,
            ),
            "pgenhip_emit_lines",
        )
        return out

    def synth_records(
        self,
        n_variants: int,
        first_variant: int = 0,
        seed: int = 0x5047454E,
        record_stride: Optional[int] = None,
        dirty_pad: bool = False,
        out: Optional[torch.Tensor] = None,
        out_offset: int = 0,
        hwe: bool = False,
    ) -> torch.Tensor:
        """Synthetic records generated on the device (bit-exact twin of the oracle's generator); ``hwe``: the
        Hardy-Weinberg value distribution of SURVEY.md §8d instead of uniform codes."""
        if record_stride is None:
            record_stride = self.record_size
        if out is None:
            out = torch.zeros(out_offset + max(n_variants, 1) * max(record_stride, 1), dtype=torch.uint8, device=self.torch_device)
        self._check_dev(out, "out")
        check(
            lib.pgenhip_synth_records(
                self._ctx, _ptr(out, out_offset), record_stride, first_variant, n_variants, seed,
                (_capi.SYNTH_DIRTY_PAD if dirty_pad else 0) | (_capi.SYNTH_HWE if hwe else 0),
            ),
            "pgenhip_synth_records",
        )
        return out

    def _check_dev(self, t: torch.Tensor, name: str) -> None:
        if not t.is_cuda or t.device.index != self.device:
            raise ValueError(f"{name} must live on cuda:{self.device}")
        if not t.is_contiguous():
            raise ValueError(f"{name} must be contiguous")
