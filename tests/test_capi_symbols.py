"""CPU leg: libpgen_hip.so loads, exports every symbol include/pgen_hip.h declares, its pure
host functions agree with the oracle, and it refuses to run without a device (no fallback)."""
import ctypes as C
import re
from pathlib import Path

import pytest
import torch

import pgen_oracle as oracle
import pgen_rs_amd
from pgen_rs_amd import _capi

REPO = Path(__file__).resolve().parent.parent


def declared_symbols():
    text = (REPO / "include" / "pgen_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pgenhip_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_what_binding_expects():
    assert declared_symbols() == sorted(_capi.PROTOTYPES.keys())


@pytest.mark.parametrize("sym", declared_symbols())
def test_symbol_exported(sym):
    handle = C.CDLL(str(_capi.LIB_PATH))
    assert getattr(handle, sym) is not None


def test_abi_version():
    assert _capi.lib.pgenhip_abi_version() == 2


def test_record_size_matches_oracle():
    for n in list(range(0, 130)) + [2504, 500000, 300, 2**31 - 1]:
        assert pgen_rs_amd.variant_record_size(n) == oracle.variant_record_size(n)


def test_record_offset_is_u64_exact():
    for v, r in [(0, 626), (17783, 626), (34359, 125000), (34360, 125000), (999_999, 125000)]:
        assert pgen_rs_amd.record_offset(v, r) == oracle.record_offset_exact(v, r) == 12 + v * r


def test_parse_header_statuses():
    good = bytes([0x6C, 0x1B, 0x02]) + (17784).to_bytes(4, "little") + (2504).to_bytes(4, "little") + b"\x40"
    assert pgen_rs_amd.parse_header(good) == (17784, 2504)
    for bad, status in [
        (b"\x6c\x1c" + good[2:], _capi.ERR_BAD_MAGIC),
        (good[:2] + b"\x10" + good[3:], _capi.ERR_BAD_MODE),
        (good[:11] + b"\x00", _capi.ERR_BAD_FLAGS),
    ]:
        with pytest.raises(pgen_rs_amd.PgenHipError) as ei:
            pgen_rs_amd.parse_header(bad)
        assert ei.value.status == status


def test_shard_range_is_the_one_partitioner():
    from pgen_rs_amd.sharding import shard_range

    b, e = C.c_uint64(), C.c_uint64()
    assert _capi.lib.pgenhip_shard_range(10, 0, 0, C.byref(b), C.byref(e)) == _capi.ERR_BAD_ARG
    assert _capi.lib.pgenhip_shard_range(10, 2, 2, C.byref(b), C.byref(e)) == _capi.ERR_BAD_ARG
    assert [shard_range(1_000_000, 8, r) for r in range(8)] == [(125_000 * r, 125_000 * (r + 1)) for r in range(8)]
    assert [shard_range(10, 3, r) for r in range(3)] == [(0, 4), (4, 7), (7, 10)]


def test_no_getenv_on_the_launch_path():
    # launch-shape knobs live in the ctx (pgenhip_tune); the library must not be steered by the process environment
    # (nor the host above it: `--filter-threads` is a flag)
    for p in list((REPO / "pgen_rs_amd" / "csrc").glob("*")) + list((REPO / "pgen_rs_amd" / "host").glob("*")):
        assert "getenv" not in p.read_text(), p


def test_null_ctx_is_refused_not_dereferenced():
    lib = _capi.lib
    assert lib.pgenhip_tune(None, _capi.KNOB_WIDE_RANGES, 8) == _capi.ERR_BAD_ARG
    assert lib.pgenhip_set_stream(None, None) == _capi.ERR_BAD_ARG
    assert lib.pgenhip_wait(None) == _capi.ERR_BAD_ARG
    assert lib.pgenhip_decode_emit(None, None, 0, None, 0, None, 0, 0) == _capi.ERR_BAD_ARG
    assert lib.pgenhip_decode_emit_at(None, None, None, 0, None, 0, 0) == _capi.ERR_BAD_ARG
    assert lib.pgenhip_destroy(None) == _capi.OK          # like free(NULL)
    assert lib.pgenhip_kept_count(None) == 0 and lib.pgenhip_gt_row_bytes(None) == 0


def test_strerror_covers_all_statuses():
    for s in range(0, -13, -1):
        assert _capi.lib.pgenhip_strerror(s) not in (None, b"unknown status")


@pytest.mark.skipif(torch.cuda.is_available(), reason="only meaningful on a box without a GPU")
def test_no_device_means_loud_failure_not_fallback():
    ctx = C.c_void_p()
    rc = _capi.lib.pgenhip_create(C.byref(ctx), 0, 100, None, 0, 0)
    assert rc == _capi.ERR_NO_DEVICE
    assert not ctx.value


def test_product_package_never_touches_oracle():
    # the product path must not import/link/execute anything under oracle/
    for p in list((REPO / "pgen_rs_amd").rglob("*.py")) + list((REPO / "pgen_rs_amd").rglob("*.hip")) + \
            list((REPO / "pgen_rs_amd").rglob("*.cpp")) + list((REPO / "pgen_rs_amd").rglob("*.h")):
        text = p.read_text()
        assert "pgen_oracle" not in text and "pgo_" not in text, p
