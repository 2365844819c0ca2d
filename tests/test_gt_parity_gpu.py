"""GPU leg: the gfx950 kernels, called through the C ABI, against the CPU oracle and the
committed golden vectors.  Bit-exact (byte/integer work): every comparison is equality of bytes.

Edge cases follow SURVEY.md §4: N % 4 in {0,1,2,3}, N < 64, N not a multiple of 64/256/512,
K = 0 / 1 / N, sparse/dense/clustered keep lists, dirty pad bits, V = 1, gapped variant lists,
unaligned output pointers, padded strides (gap bytes must stay untouched).
"""
import hashlib

import numpy as np
import pytest
import torch

import pgen_oracle as oracle
import pgen_rs_amd
from helpers import case_names, load_case, sha_table
from pgen_rs_amd import _capi

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
SENTINEL = 0xA5


def kernels_for(subset: bool, dense: bool, n: int = 0, k: int = 0):
    ks = [_capi.KERNEL_AUTO, _capi.KERNEL_ROWS]
    if not subset and dense and n >= 8:
        ks.append(_capi.KERNEL_FLAT)
    if not subset and dense and n >= 1024:
        ks.append(_capi.KERNEL_WIDE)
    if subset and n >= 61:  # the scan kernel needs records of >= 16 bytes
        ks.append(_capi.KERNEL_SCAN)
    if dense and 61 <= n <= 4096 and (k >= 1 if subset else True):
        ks.append(_capi.KERNEL_PICK)  # short records: kept subset through the table, or all samples (identity)
    if subset and n >= 61 and 1 <= k <= 16384:
        ks.append(_capi.KERNEL_ROWPICK)  # one wave per row (any strides)
    return ks


def run_engine(recs_np, v, n, kept=None, kernel=_capi.KERNEL_AUTO, record_stride=None, out_stride=None,
               variant_idx=None, out_offset=0, records_offset=0, tune=None):
    """Runs the HIP path; returns the whole output buffer (sentinel-filled where untouched).
    `tune`: {knob: value} applied to the context first (pgenhip_tune: small grids, forced kernel bands)."""
    with pgen_rs_amd.GtEngine(n, kept_idx=kept, device=0) as eng:
        for knob, value in (tune or {}).items():
            eng.tune(knob, value)
        k = eng.kept_count
        if out_stride is None:
            out_stride = 4 * k + 1
        rec_t = torch.from_numpy(np.array(recs_np, dtype=np.uint8, copy=True)).to(DEV)
        total = out_offset + max(v, 1) * out_stride + 64
        out = torch.full((total,), SENTINEL, dtype=torch.uint8, device=DEV)
        vidx_t = None
        if variant_idx is not None:
            vidx_t = torch.tensor(np.asarray(variant_idx, dtype=np.int64), dtype=torch.int32, device=DEV)
        eng.decode_emit(rec_t, v, record_stride=record_stride, variant_idx=vidx_t, out=out, out_stride=out_stride,
                        kernel=kernel, out_offset=out_offset, records_offset=records_offset)
        eng.wait()
        return out.cpu().numpy(), k


def expect_buffer(want_rows, v, k, out_stride, out_offset, total):
    buf = np.full(total, SENTINEL, dtype=np.uint8)
    row = 4 * k + 1
    for j in range(v):
        buf[out_offset + j * out_stride : out_offset + j * out_stride + row] = want_rows[j]
    return buf


@pytest.mark.parametrize("name", case_names())
def test_golden_cases(name):
    v, n, recs, kept, gt = load_case(name)
    for kern in kernels_for(kept is not None, True, n):
        got, k = run_engine(recs.reshape(-1), v, n, kept=kept, kernel=kern)
        assert bytes(got[: gt.size]) == bytes(gt), f"kernel {kern}"
        assert (got[gt.size :] == SENTINEL).all(), f"kernel {kern} wrote past the end"


@pytest.mark.parametrize("name", sorted(sha_table().keys()))
def test_golden_sha_with_device_synth(name):
    spec = sha_table()[name]
    n, v = spec["sample_count"], spec["n_variants"]
    kept = oracle.synth_keep(n, spec["keep_seed"], spec["keep_modulus"]) if spec["keep_modulus"] else None
    with pgen_rs_amd.GtEngine(n, kept_idx=kept, device=0) as eng:
        recs = eng.synth_records(v, first_variant=spec["first_variant"], seed=spec["seed"], dirty_pad=spec["dirty_pad"],
                                 hwe=spec["distribution"] == "hwe")
        eng.wait()
        assert hashlib.sha256(recs.cpu().numpy().tobytes()).hexdigest() == spec["records_sha256"]
        ks = kernels_for(kept is not None, True, n)
        if kept is None and 8 <= n <= 1915:
            ks.append(_capi.KERNEL_RUNS)
        for kern in ks:
            out = eng.decode_emit(recs, v, kernel=kern)
            eng.wait()
            assert out.numel() == spec["gt_bytes"]
            assert hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest() == spec["gt_sha256"], f"kernel {kern}"


@pytest.mark.parametrize("n", [0, 1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 63, 64, 65, 66, 67, 255, 256, 257, 511, 513, 1023, 1024, 1025, 2047, 2048, 2049, 2051, 2504, 4093, 4095, 4096, 4097, 4099, 8191, 8193, 16385, 70001])
def test_all_samples_vs_oracle(n):
    rng = np.random.default_rng(n + 1)
    v = 11
    r = oracle.variant_record_size(n)
    recs = rng.integers(0, 256, size=max(v * r, 1), dtype=np.uint8)  # dirty pad bits included
    want = oracle.decode_emit(recs, v, n).reshape(v, -1)
    for kern in kernels_for(False, True, n):
        got, k = run_engine(recs, v, n, kernel=kern)
        assert k == n
        exp = expect_buffer(want, v, n, 4 * n + 1, 0, got.size)
        assert (got == exp).all(), f"n={n} kernel {kern}"


@pytest.mark.parametrize("out_offset", range(0, 17))
def test_unaligned_output_pointer(out_offset):
    rng = np.random.default_rng(100 + out_offset)
    n, v = ((131, 6) if out_offset % 2 else (1031, 5)) if out_offset % 3 else (2053, 9)
    r = oracle.variant_record_size(n)
    recs = rng.integers(0, 256, size=v * r, dtype=np.uint8)
    want = oracle.decode_emit(recs, v, n).reshape(v, -1)
    for kern in kernels_for(False, True, n):
        got, _ = run_engine(recs, v, n, kernel=kern, out_offset=out_offset)
        exp = expect_buffer(want, v, n, 4 * n + 1, out_offset, got.size)
        assert (got == exp).all(), f"kernel {kern}"


@pytest.mark.parametrize("pad", [1, 2, 3, 5, 15, 16, 17, 127])
def test_padded_strides_leave_gaps_untouched(pad):
    rng = np.random.default_rng(200 + pad)
    n, v = 203, 7
    r = oracle.variant_record_size(n)
    rstride = r + pad
    recs = rng.integers(0, 256, size=v * rstride + 3, dtype=np.uint8)
    dense = np.concatenate([recs[3 + i * rstride : 3 + i * rstride + r] for i in range(v)])
    want = oracle.decode_emit(dense, v, n).reshape(v, -1)
    ostride = 4 * n + 1 + pad
    got, _ = run_engine(recs, v, n, record_stride=rstride, out_stride=ostride, records_offset=3)
    exp = expect_buffer(want, v, n, ostride, 0, got.size)
    assert (got == exp).all()


def test_variant_index_gather():
    rng = np.random.default_rng(7)
    n, vfile = 1001, 50
    r = oracle.variant_record_size(n)
    recs = rng.integers(0, 256, size=vfile * r, dtype=np.uint8)
    vidx = [49, 0, 7, 8, 9, 30, 30, 2]
    want = oracle.decode_emit(recs, len(vidx), n, variant_idx=vidx).reshape(len(vidx), -1)
    got, _ = run_engine(recs, len(vidx), n, variant_idx=vidx)
    exp = expect_buffer(want, len(vidx), n, 4 * n + 1, 0, got.size)
    assert (got == exp).all()


def keep_lists(n, rng):
    yield "k0", np.array([], dtype=np.uint32)
    yield "first", np.array([0], dtype=np.uint32)
    yield "last", np.array([n - 1], dtype=np.uint32)
    yield "all_as_list", np.arange(n, dtype=np.uint32)
    yield "every2", np.arange(0, n, 2, dtype=np.uint32)
    yield "every3_off1", np.arange(1, n, 3, dtype=np.uint32)
    yield "cluster", np.arange(n // 3, min(n, n // 3 + 70), dtype=np.uint32)
    yield "sparse1pct", np.sort(rng.choice(n, size=max(1, n // 100), replace=False)).astype(np.uint32)
    yield "dense_holes", np.setdiff1d(np.arange(n), rng.choice(n, size=max(1, n // 50), replace=False)).astype(np.uint32)
    yield "two_ends", np.array([0, n - 1], dtype=np.uint32) if n > 1 else np.array([0], dtype=np.uint32)


@pytest.mark.parametrize("n", [1, 5, 63, 64, 65, 130, 257, 1000, 2504, 4097, 20011])
def test_kept_subsets_vs_oracle(n):
    rng = np.random.default_rng(300 + n)
    v = 5
    r = oracle.variant_record_size(n)
    recs = rng.integers(0, 256, size=v * r, dtype=np.uint8)
    for label, kept in keep_lists(n, rng):
        want = oracle.decode_emit(recs, v, n, kept_idx=kept).reshape(v, -1)
        for kern in kernels_for(True, True, n, int(kept.size)):
            got, k = run_engine(recs, v, n, kept=kept, kernel=kern)
            assert k == kept.size
            exp = expect_buffer(want, v, k, 4 * k + 1, 0, got.size)
            assert (got == exp).all(), f"n={n} keep={label} kernel {kern}"


def test_subset_with_strides_offsets_and_gather():
    rng = np.random.default_rng(9)
    n, vfile = 777, 20
    r = oracle.variant_record_size(n)
    rstride = r + 9
    recs = rng.integers(0, 256, size=vfile * rstride + 5, dtype=np.uint8)
    kept = np.arange(2, n, 5, dtype=np.uint32)
    vidx = [3, 19, 4, 0]
    dense = np.concatenate([recs[5 + i * rstride : 5 + i * rstride + r] for i in range(vfile)])
    want = oracle.decode_emit(dense, len(vidx), n, kept_idx=kept, variant_idx=vidx).reshape(len(vidx), -1)
    k = kept.size
    for kern in kernels_for(True, False, n, k):
        got, _ = run_engine(recs, len(vidx), n, kept=kept, kernel=kern, record_stride=rstride, out_stride=4 * k + 1 + 6,
                            variant_idx=vidx, out_offset=3, records_offset=5)
        exp = expect_buffer(want, len(vidx), k, 4 * k + 1 + 6, 3, got.size)
        assert (got == exp).all(), f"kernel {kern}"


@pytest.mark.parametrize("per_cu", [1, 2])
def test_scan_kernels_many_rows_per_wave(per_cu):
    """The segment kernel with many rows per wave (1 or 2 blocks per CU): register double-buffering over many
    rows, segments with no kept sample at the front and at the back (the last segment then only owes the
    '\n'), a locally dense mask and a gapped variant list."""
    tune = {_capi.KNOB_SCAN_BLOCKS_PER_CU: per_cu}  # few blocks -> ~8-25 rows per wave
    n = 40000  # three 16 384-sample segments, the last one partial
    r = oracle.variant_record_size(n)
    rng = np.random.default_rng(77)
    masks = {
        "1pct": np.sort(rng.choice(n, size=n // 100, replace=False)),
        "5pct": np.sort(rng.choice(n, size=n * 5 // 100, replace=False)),      # ~820 per segment
        "11pct": np.sort(rng.choice(n, size=n * 11 // 100, replace=False)),    # ~1 800 per segment
        "front_empty": np.sort(rng.choice(np.arange(16384, n), size=300, replace=False)),
        "back_empty": np.sort(rng.choice(np.arange(0, 32768), size=300, replace=False)),
        "only_middle": np.sort(rng.choice(np.arange(16384, 32768), size=500, replace=False)),
        "locally_dense": np.concatenate([np.arange(100, 5100), np.sort(rng.choice(np.arange(20000, n), size=50, replace=False))]),
    }
    for label, kept in masks.items():
        kept = kept.astype(np.uint32)
        for v, vidx in ((6001, None), (1999, "gapped")):
            v_file = v if vidx is None else v * 2
            recs = rng.integers(0, 256, size=v_file * r, dtype=np.uint8)
            idx = None if vidx is None else np.sort(rng.choice(v_file, size=v, replace=False))
            want = oracle.decode_emit(recs, v, n, kept_idx=kept, variant_idx=idx).reshape(v, -1)
            got, k = run_engine(recs, v, n, kept=kept, kernel=_capi.KERNEL_SCAN, variant_idx=idx, tune=tune)
            exp = expect_buffer(want, v, k, 4 * k + 1, 0, got.size)
            assert (got == exp).all(), f"mask {label} v={v} per_cu={per_cu}"


@pytest.mark.parametrize("n,k", [(40_000, 240), (40_000, 241), (40_000, 401), (70_001, 1_915), (70_001, 1_916), (100_003, 4_001), (147_457, 6_002), (147_457, 6_003)])
def test_two_pass_path_for_sparse_keeps(n, k):
    """Sparse keeps on long records take TWO passes in AUTO (compact records of K samples, then the all-samples kernels on
    them): K % 4 in {0,1,2,3} (compact-byte ownership at every segment seam, the ranks behind a slice fetched from later
    segments), K on both sides of the RUNS / row-item hand-over of the second pass, kept samples clustered at segment
    ends, chunks of 1, 7 and all rows, a gapped variant list, GT segments and full lines; against the oracle and against the
    single-pass segment kernel."""
    rng = np.random.default_rng(n + k)
    r = oracle.variant_record_size(n)
    v = 61
    seam = np.array([s for s in (16383, 16384, 16385, 32767, 32768, 49151, 49152) if s < n])
    kept = np.unique(np.concatenate([[0, n - 1], seam, rng.choice(n, size=k, replace=False)]))[:k].astype(np.uint32)
    if kept.size < k:
        kept = np.unique(np.concatenate([kept, rng.choice(n, size=k, replace=False)]))[:k].astype(np.uint32)
    assert kept.size == k and k * 170 >= n and k * 22 <= n  # inside the two-pass band
    v_file = v + 13
    recs = rng.integers(0, 256, size=v_file * r, dtype=np.uint8)
    vidx = np.sort(rng.choice(v_file, size=v, replace=False))
    want = oracle.decode_emit(recs, v, n, kept_idx=kept, variant_idx=vidx).reshape(v, -1)
    for chunk in (0, 1, 7):
        tune = {_capi.KNOB_SCAN_CHUNK_ROWS: chunk}
        for use_vidx in (True, False):
            w = want if use_vidx else oracle.decode_emit(recs, v, n, kept_idx=kept).reshape(v, -1)
            got, kk = run_engine(recs, v, n, kept=kept, variant_idx=vidx if use_vidx else None, out_offset=3, tune=tune)
            exp = expect_buffer(w, v, kk, 4 * kk + 1, 3, got.size)
            assert (got == exp).all(), f"two-pass chunk={chunk} vidx={use_vidx}"
    got1, kk = run_engine(recs, v, n, kept=kept, variant_idx=vidx, out_offset=3, tune={_capi.KNOB_SCAN_TWO_PASS: -1})
    assert (got1 == expect_buffer(want, v, kk, 4 * kk + 1, 3, got1.size)).all(), "single pass"
    # full lines
    prefixes = [bytes(rng.integers(33, 127, size=int(rng.integers(0, 40)) if i % 3 else 0, dtype=np.uint8)) for i in range(v)]
    blob = np.frombuffer(b"".join(prefixes) + b"!", dtype=np.uint8)
    poff = np.cumsum([0] + [len(q) for q in prefixes]).astype(np.int64)
    loff = np.cumsum([0] + [len(q) + 4 * k + 1 for q in prefixes]).astype(np.int64)
    wl = oracle.emit_lines(recs, v, n, blob, poff.astype(np.uint64), loff.astype(np.uint64), kept_idx=kept, variant_idx=vidx)
    with pgen_rs_amd.GtEngine(n, kept_idx=kept, device=0) as eng:
        for chunk in (0, 5):
            eng.tune(_capi.KNOB_SCAN_CHUNK_ROWS, chunk)
            out = torch.full((2 + int(loff[-1]) + 48,), SENTINEL, dtype=torch.uint8, device=DEV)
            eng.emit_lines(torch.from_numpy(recs).to(DEV), v, torch.from_numpy(blob.copy()).to(DEV), torch.from_numpy(poff).to(DEV),
                           torch.from_numpy(loff).to(DEV), 40, out[2:], variant_idx=torch.tensor(vidx, dtype=torch.int32, device=DEV))
            eng.wait()
            got = out.cpu().numpy()
            assert (got[:2] == SENTINEL).all() and (got[2 + wl.size :] == SENTINEL).all()
            assert bytes(got[2 : 2 + wl.size]) == wl.tobytes(), f"lines, chunk={chunk}"


@pytest.mark.parametrize("n", [16384, 16385, 49152, 49153, 70001, 98304, 100003, 147457])
def test_sparse_subsets_segment_triples(n):
    """Sparse keeps where the number of 16 384-sample segments is 1..10, i.e. the last block of the
    last block of a row group owns a partial segment and the record ends in any of its
    tiles; kept samples include the very first and the very last sample."""
    rng = np.random.default_rng(900 + n)
    v = 41
    r = oracle.variant_record_size(n)
    recs = rng.integers(0, 256, size=v * r, dtype=np.uint8)
    kept = np.unique(np.concatenate([[0, n - 1], rng.choice(n, size=max(2, n // 150), replace=False)])).astype(np.uint32)
    want = oracle.decode_emit(recs, v, n, kept_idx=kept).reshape(v, -1)
    for kern in (_capi.KERNEL_AUTO, _capi.KERNEL_SCAN):
        got, k = run_engine(recs, v, n, kept=kept, kernel=kern)
        exp = expect_buffer(want, v, k, 4 * k + 1, 0, got.size)
        assert (got == exp).all(), f"n={n} kernel {kern}"


@pytest.mark.parametrize("n", [61, 64, 100, 257, 1000, 2504, 4093, 4096])
def test_pick_kernel_short_records(n):
    """Short-record subset kernel (N <= 4096): K from 1 to N-1 (K = 1, 2, 3: rows of 5 / 9 / 13 bytes, a 16-byte chunk spans up
    to four rows — the README's `IID == "NA20900"` example; rows of 17 bytes up: at most two), batches that
    end mid-way (V not a multiple of the batch), unaligned output pointers (the run's head and tail
    edges), a gapped variant list and a padded record stride; sentinel bytes stay untouched."""
    rng = np.random.default_rng(1200 + n)
    r = oracle.variant_record_size(n)
    masks = {
        "k1": np.array([n * 2 // 3]),
        "k1_last": np.array([n - 1]),
        "k2": np.sort(rng.choice(n, size=2, replace=False)),
        "k3_ends": np.array([0, n // 2, n - 1]),
        "k4": np.sort(rng.choice(n, size=4, replace=False)),
        "k5_ends": np.unique(np.concatenate([[0, n - 1], rng.choice(n, size=3, replace=False)])),
        "1pct": np.sort(rng.choice(n, size=max(4, n // 100), replace=False)),
        "half": np.sort(rng.choice(n, size=n // 2, replace=False)),
        "all_but_one": np.setdiff1d(np.arange(n), [n // 3]),
    }
    for label, kept in masks.items():
        kept = kept.astype(np.uint32)
        for v, out_offset, gapped in ((1, 0, False), (7, 3, False), (1031, 0, False), (397, 9, True)):
            v_file = v * 2 if gapped else v
            rstride = r + 5 if gapped else r
            recs = rng.integers(0, 256, size=v_file * rstride + 3, dtype=np.uint8)
            vidx = np.sort(rng.choice(v_file, size=v, replace=False)) if gapped else None
            dense = np.concatenate([recs[3 + i * rstride : 3 + i * rstride + r] for i in range(v_file)])
            want = oracle.decode_emit(dense, v, n, kept_idx=kept, variant_idx=vidx).reshape(v, -1)
            got, k = run_engine(recs, v, n, kept=kept, kernel=_capi.KERNEL_PICK, record_stride=rstride, variant_idx=vidx,
                                out_offset=out_offset, records_offset=3)
            exp = expect_buffer(want, v, k, 4 * k + 1, out_offset, got.size)
            if not (got == exp).all():
                bad = np.flatnonzero(got != exp)
                raise AssertionError(f"n={n} mask={label} v={v} off={out_offset}: {bad.size} bytes differ, first at {bad[:6]}")


@pytest.mark.parametrize("n", [8, 9, 10, 11, 33, 61, 100, 255, 256, 300, 301, 302, 303, 500, 1000, 1024, 1399, 1915, 1916, 2504, 3831])
def test_runs_kernel_short_rows(n):
    """RUNS mode of the stream kernel (short rows: a work item is a run of consecutive rows; the '\n' chunks of a
    run are written in a separate pass).  N from 8 (33-byte rows) up to the largest N with two rows per item,
    incl. the reference's own dataset shape N = 300 and N % 4 in {0,1,2,3}; V = 1, V < one run, V not a multiple
    of the run, many runs per block (ring re-use, queue stealing); unaligned output AND record pointers (every
    phase of the first chunk and of the wide load); forced short runs (2, 3 rows); sentinels around the output."""
    rng = np.random.default_rng(3300 + n)
    r = oracle.variant_record_size(n)
    for v, out_off, rec_off, rows_knob in ((1, 0, 0, 0), (2, 1, 3, 0), (7, 15, 1, 2), (64, 16, 15, 3), (1031, 5, 7, 0), (20_011, 127, 0, 0)):
        recs = rng.integers(0, 256, size=rec_off + v * r, dtype=np.uint8)
        want = oracle.decode_emit(recs[rec_off:], v, n).reshape(v, -1)
        tune = {_capi.KNOB_RUNS_ROWS: rows_knob}
        if v > 10_000:
            tune[_capi.KNOB_WIDE_BLOCKS_PER_CU] = 1
        got, _ = run_engine(recs, v, n, kernel=_capi.KERNEL_RUNS, out_offset=out_off, records_offset=rec_off, tune=tune)
        exp = expect_buffer(want, v, n, 4 * n + 1, out_off, got.size)
        if not (got == exp).all():
            bad = np.flatnonzero(got != exp)
            raise AssertionError(f"n={n} v={v} out_off={out_off} rec_off={rec_off} rows={rows_knob}: {bad.size} bytes differ, first at {bad[:8]}")


def test_runs_kernel_refuses_what_it_cannot_do():
    with pgen_rs_amd.GtEngine(300, device=0) as eng:
        recs = torch.zeros(10 * 80, dtype=torch.uint8, device=DEV)
        out = torch.zeros(10 * 1201, dtype=torch.uint8, device=DEV)
        with pytest.raises(pgen_rs_amd.PgenHipError):   # padded record stride: the run's records are not contiguous
            eng.decode_emit(recs, 10, record_stride=80, out=out, kernel=_capi.KERNEL_RUNS)
        vidx = torch.arange(10, dtype=torch.int32, device=DEV)
        with pytest.raises(pgen_rs_amd.PgenHipError):   # variant gather
            eng.decode_emit(recs, 10, variant_idx=vidx, out=out, kernel=_capi.KERNEL_RUNS)
    with pgen_rs_amd.GtEngine(5000, device=0) as eng:   # a row of 20 001 bytes does not fit one item (N <= 3 831)
        with pytest.raises(pgen_rs_amd.PgenHipError):
            eng.decode_emit(torch.zeros(1250 * 4, dtype=torch.uint8, device=DEV), 4, kernel=_capi.KERNEL_RUNS)


def test_work_queue_heads_alternate_across_launches():
    """One context, many launches: the last block of a work-queue launch re-zeroes its block of queue heads and
    every launch takes the next block of the ring (more launches than the ring has blocks).  Interleave
    work-queue launches with launches of kernels that do not use the queue and check every output."""
    n, v = 2504, 3001
    rng = np.random.default_rng(4242)
    r = oracle.variant_record_size(n)
    recs = rng.integers(0, 256, size=v * r, dtype=np.uint8)
    want = oracle.decode_emit(recs, v, n).tobytes()
    with pgen_rs_amd.GtEngine(n, device=0) as eng:
        d_recs = torch.from_numpy(recs).to(DEV)
        plan = [_capi.KERNEL_WIDE, _capi.KERNEL_WIDE, _capi.KERNEL_ROWS, _capi.KERNEL_WIDE, _capi.KERNEL_FLAT,
                _capi.KERNEL_WIDE, _capi.KERNEL_PICK, _capi.KERNEL_WIDE, _capi.KERNEL_WIDE] * 5  # 45 launches > 2 x the ring
        for step, kern in enumerate(plan):
            out = torch.full((v * (4 * n + 1),), SENTINEL, dtype=torch.uint8, device=DEV)
            eng.decode_emit(d_recs, v, out=out, kernel=kern)
            eng.wait()
            assert out.cpu().numpy().tobytes() == want, f"launch {step} (kernel {kern})"


def test_launches_of_one_ctx_on_different_streams_overlap():
    """include/pgen_hip.h "Streams": a ctx may be moved between streams between back-to-back launches and the
    launches may overlap — each takes its own block of work-queue counters.  Queue 12 work-queue launches of
    ONE ctx round-robin on three streams without waiting in between (few blocks per CU so that they really
    run side by side), then compare every output with the oracle."""
    n, v = 2504, 6001
    r = oracle.variant_record_size(n)
    with pgen_rs_amd.GtEngine(n, device=0) as eng:
        eng.tune(_capi.KNOB_WIDE_BLOCKS_PER_CU, 1)
        streams = [torch.cuda.Stream(device=DEV) for _ in range(3)]
        recs, outs = [], []
        for i in range(12):
            recs.append(eng.synth_records(v, first_variant=1000 * i))
            outs.append(torch.full((v * (4 * n + 1),), SENTINEL, dtype=torch.uint8, device=DEV))
        torch.cuda.synchronize()
        for i in range(12):
            eng.use_stream(streams[i % 3])
            eng.decode_emit(recs[i], v, out=outs[i], kernel=_capi.KERNEL_WIDE)
        torch.cuda.synchronize()
        eng.use_torch_stream()
        for i in range(12):
            host = recs[i][: v * r].cpu().numpy()
            assert outs[i].cpu().numpy().tobytes() == oracle.decode_emit(host, v, n).tobytes(), f"launch {i}"
        # and the ctx is still good for an ordinary launch afterwards
        out = eng.decode_emit(recs[0], v)
        eng.wait()
        assert out.cpu().numpy().tobytes() == oracle.decode_emit(recs[0][: v * r].cpu().numpy(), v, n).tobytes()


@pytest.mark.parametrize("n,k,v", [(40_000, 400, 9_001), (70_001, 1_900, 8_200), (16_385, 300, 8_193), (100_003, 4_001, 8_500)])
@pytest.mark.parametrize("mode", ["gt", "lines", "gather_padded"])
def test_row_owner_kernel_sparse_keeps_many_rows(n, k, v, mode):
    """BASELINE configs[4]'s band with enough rows for every resident wave (AUTO -> gt_rowpick.hip: one wave per row, the row's
    compact record assembled in LDS segment by segment — bytes that straddle two segments, empty segments, the record's tail
    tile — and its text written in one go): GT segments, full lines (the kernel writes the prefixes itself), and a gathered,
    padded layout; AUTO and the forced kernel agree with the oracle byte for byte, sentinels untouched."""
    rng = np.random.default_rng(n + k)
    kept = np.sort(rng.choice(n, size=k, replace=False)).astype(np.uint32)
    if n == 70_001:
        kept = np.sort(np.concatenate([rng.choice(20_000, size=k - 3, replace=False), [n - 1, n - 2, 49_152]])).astype(np.uint32)  # empty segments 1 and 2, the last sample
        kept = np.unique(kept)
        k = len(kept)
    r = oracle.variant_record_size(n)
    with pgen_rs_amd.GtEngine(n, kept_idx=kept, device=0) as eng:
        if mode == "gather_padded":
            stride = r + 7
            v_file = v + 50
            base = eng.synth_records(v_file, first_variant=3, record_stride=stride)
            vidx = np.sort(rng.choice(v_file, size=v, replace=False)).astype(np.int32)
            host = base.cpu().numpy()
            dense = np.concatenate([host[i * stride : i * stride + r] for i in vidx])
            want = oracle.decode_emit(dense, v, n, kept_idx=kept).reshape(v, -1)
            ostride = 4 * k + 1 + 5
            for kern in (_capi.KERNEL_ROWPICK,):
                out = torch.full((v * ostride + 16,), SENTINEL, dtype=torch.uint8, device=DEV)
                eng.decode_emit(base, v, record_stride=stride, variant_idx=torch.from_numpy(vidx).to(DEV), out=out, out_stride=ostride, kernel=kern)
                eng.wait()
                got = out.cpu().numpy()
                exp = expect_buffer(want, v, k, ostride, 0, got.size)
                assert (got == exp).all(), f"kernel {kern}"
            return
        recs = eng.synth_records(v, first_variant=17)
        host = recs[: v * r].cpu().numpy()
        if mode == "gt":
            want = oracle.decode_emit(host, v, n, kept_idx=kept).tobytes()
            for kern in (_capi.KERNEL_AUTO, _capi.KERNEL_ROWPICK):
                out = torch.full((5 + v * (4 * k + 1) + 16,), SENTINEL, dtype=torch.uint8, device=DEV)
                for mode_knob in ((1, -1, 2) if kern == _capi.KERNEL_AUTO else (1,)):   # AUTO: row-owner compact pass, segment compact pass, row-owner single pass
                    eng.tune(_capi.KNOB_SCAN_ROWPICK, mode_knob)
                    out.fill_(SENTINEL)
                    eng.decode_emit(recs, v, out=out, kernel=kern, out_offset=5)
                    eng.wait()
                    got = out.cpu().numpy()
                    assert (got[:5] == SENTINEL).all() and (got[5 + len(want) :] == SENTINEL).all(), f"kernel {kern} knob {mode_knob} wrote outside"
                    assert got[5 : 5 + len(want)].tobytes() == want, f"kernel {kern} knob {mode_knob}"
                eng.tune(_capi.KNOB_SCAN_ROWPICK, 1)
        else:
            plen = rng.integers(0, 70, size=v).astype(np.int64)
            poff = np.concatenate([[0], np.cumsum(plen)]).astype(np.int64)
            loff = np.concatenate([[0], np.cumsum(plen + 4 * k + 1)]).astype(np.int64)
            blob = rng.integers(65, 91, size=int(poff[-1]) + 1, dtype=np.uint8)
            want = oracle.emit_lines(host, v, n, blob, poff.astype(np.uint64), loff.astype(np.uint64), kept_idx=kept).tobytes()
            d_blob, d_poff, d_loff = (torch.from_numpy(x).to(DEV) for x in (blob, poff, loff))
            for kern in (_capi.KERNEL_AUTO, _capi.KERNEL_ROWPICK):
                out = torch.full((3 + int(loff[-1]) + 16,), SENTINEL, dtype=torch.uint8, device=DEV)
                eng.emit_lines(recs, v, d_blob, d_poff, d_loff, int(plen.max()), out[3:], kernel=kern)
                eng.wait()
                got = out.cpu().numpy()
                assert (got[:3] == SENTINEL).all() and (got[3 + len(want) :] == SENTINEL).all(), f"kernel {kern} wrote outside"
                assert got[3 : 3 + len(want)].tobytes() == want, f"kernel {kern}"


@pytest.mark.parametrize("seed", range(6))
def test_row_owner_kernel_randomized_forced(seed):
    """The row-owner kernel FORCED (PGENHIP_KERNEL_ROWPICK) on random shapes AUTO would never give it: N from 61 to 200 000 (one to
    thirteen segments, records that end anywhere in a tile), K from 1 to 16 384 in every arrangement (dense runs, empty segments,
    first / last sample), few rows, gathered and padded records, padded output strides, unaligned pointers, GT segments and
    full lines with prefixes of 0 .. 60 bytes; sentinels around everything it may not touch."""
    rng = np.random.default_rng(4400 + seed)
    for case_i in range(10):
        n = int(rng.choice([61, 64, 300, 4096, 4097, 16384, 16385, 32768, 49153, 70001, 200_000])) if rng.random() < 0.7 else int(rng.integers(61, 120_000))
        v = int(rng.choice([1, 2, 5, 33, 257]))
        style = rng.choice(["sparse", "dense_run", "tiny", "ends", "modulus"])
        if style == "sparse":
            kept = np.sort(rng.choice(n, size=max(1, min(16384, int(n * rng.uniform(0.001, 0.05)))), replace=False))
        elif style == "dense_run":
            a0 = int(rng.integers(0, n - 1))
            kept = np.arange(a0, min(n, a0 + int(rng.integers(1, min(16384, n) + 1))))
        elif style == "tiny":
            kept = np.sort(rng.choice(n, size=int(rng.integers(1, 4)), replace=False))
        elif style == "ends":
            kept = np.unique(np.array([0, n - 1, n // 2, min(n - 1, 16383), min(n - 1, 16384)]))
        else:
            kept = np.arange(int(rng.integers(0, 7)), n, int(rng.integers(max(2, n // 16000 + 1), 200)))[:16384]
        kept = kept.astype(np.uint32)
        k = int(kept.size)
        r = oracle.variant_record_size(n)
        gather = bool(rng.random() < 0.4)
        rstride = r + (int(rng.choice([0, 1, 7, 16])) if gather else 0)
        v_file = 2 * v if gather else v
        rec_off = int(rng.integers(0, 17))
        recs = rng.integers(0, 256, size=rec_off + v_file * rstride + 16, dtype=np.uint8)
        vidx = rng.integers(0, v_file, size=v).astype(np.uint32) if gather else None
        dense = np.concatenate([recs[rec_off + i * rstride : rec_off + i * rstride + r] for i in range(v_file)])
        lines = bool(rng.random() < 0.5)
        tag = f"seed={seed} case={case_i} n={n} v={v} k={k} style={style} gather={gather} rstride={rstride} lines={lines}"
        with pgen_rs_amd.GtEngine(n, kept_idx=kept, device=0) as eng:
            rec_t = torch.from_numpy(recs).to(DEV)
            vidx_t = None if vidx is None else torch.tensor(vidx.astype(np.int64), dtype=torch.int32, device=DEV)
            out_off = int(rng.integers(0, 40))
            if lines:
                plens = rng.integers(0, 61, size=v)
                prefixes = [bytes(rng.integers(33, 127, size=int(q), dtype=np.uint8)) for q in plens]
                blob = np.frombuffer(b"".join(prefixes) + b"!", dtype=np.uint8)
                poff = np.cumsum([0] + [len(q) for q in prefixes]).astype(np.int64)
                loff = np.cumsum([0] + [len(q) + 4 * k + 1 for q in prefixes]).astype(np.int64)
                want = oracle.emit_lines(dense, v, n, blob, poff.astype(np.uint64), loff.astype(np.uint64), kept_idx=kept, variant_idx=vidx)
                out = torch.full((out_off + int(loff[-1]) + 32,), SENTINEL, dtype=torch.uint8, device=DEV)
                eng.emit_lines(rec_t, v, torch.from_numpy(blob.copy()).to(DEV), torch.from_numpy(poff).to(DEV), torch.from_numpy(loff).to(DEV), 60, out[out_off:],
                               record_stride=rstride, variant_idx=vidx_t, kernel=_capi.KERNEL_ROWPICK, records_offset=rec_off)
                eng.wait()
                got = out.cpu().numpy()
                assert (got[:out_off] == SENTINEL).all() and (got[out_off + want.size :] == SENTINEL).all(), tag + ": wrote outside the lines"
                assert got[out_off : out_off + want.size].tobytes() == want.tobytes(), tag
            else:
                ostride = 4 * k + 1 + int(rng.choice([0, 0, 3, 16]))
                want = oracle.decode_emit(dense, v, n, kept_idx=kept, variant_idx=vidx).reshape(v, -1)
                out = torch.full((out_off + v * ostride + 32,), SENTINEL, dtype=torch.uint8, device=DEV)
                eng.decode_emit(rec_t, v, record_stride=rstride, variant_idx=vidx_t, out=out, out_stride=ostride, kernel=_capi.KERNEL_ROWPICK,
                                out_offset=out_off, records_offset=rec_off)
                eng.wait()
                got = out.cpu().numpy()
                exp = expect_buffer(want, v, k, ostride, out_off, got.size)
                assert (got == exp).all(), tag


@pytest.mark.parametrize("lines", [False, True])
def test_two_pass_launches_of_one_ctx_on_different_streams_overlap(lines):
    """The same promise on the TWO-PASS path (sparse keeps on long records, BASELINE configs[4]'s band): pass 1 parks a
    chunk's compact records in ctx scratch, pass 2 reads them back.  Every launch in flight has its own slice of that
    scratch (round 2 had ONE per ctx: overlapping launches on two streams read each other's records — ADVICE r2 medium).
    9 launches of ONE ctx round-robin on three streams, small chunks (several chunk rounds per launch) and few blocks per
    CU so that the launches really interleave; every output against the oracle."""
    n, v = 40_000, 1501
    kept = np.sort(np.random.default_rng(77).choice(n, size=n // 100 if not lines else 1100, replace=False)).astype(np.uint32)
    r = oracle.variant_record_size(n)
    k = len(kept)
    with pgen_rs_amd.GtEngine(n, kept_idx=kept, device=0) as eng:
        eng.tune(_capi.KNOB_SCAN_CHUNK_ROWS, 200)
        eng.tune(_capi.KNOB_SCAN_BLOCKS_PER_CU, 1)
        eng.tune(_capi.KNOB_WIDE_BLOCKS_PER_CU, 1)
        streams = [torch.cuda.Stream(device=DEV) for _ in range(3)]
        recs, outs = [], []
        if lines:
            plen = np.random.default_rng(5).integers(3, 60, size=v).astype(np.int64)
            poff = np.concatenate([[0], np.cumsum(plen)]).astype(np.int64)
            loff = np.concatenate([[0], np.cumsum(plen + 4 * k + 1)]).astype(np.int64)
            blob = np.random.default_rng(6).integers(65, 91, size=int(poff[-1]), dtype=np.uint8)
            d_blob, d_poff, d_loff = (torch.from_numpy(x).to(DEV) for x in (blob, poff, loff))
            out_bytes = int(loff[-1])
        else:
            out_bytes = v * (4 * k + 1)
        for i in range(9):
            recs.append(eng.synth_records(v, first_variant=777 * i))
            outs.append(torch.full((out_bytes,), SENTINEL, dtype=torch.uint8, device=DEV))
        torch.cuda.synchronize()
        for i in range(9):
            eng.use_stream(streams[i % 3])
            if lines:
                eng.emit_lines(recs[i], v, d_blob, d_poff, d_loff, int(plen.max()), outs[i])
            else:
                eng.decode_emit(recs[i], v, out=outs[i])
        torch.cuda.synchronize()
        eng.use_torch_stream()
        for i in range(9):
            host = recs[i][: v * r].cpu().numpy()
            if lines:
                want = oracle.emit_lines(host, v, n, blob, poff.astype(np.uint64), loff.astype(np.uint64), kept_idx=kept).tobytes()
            else:
                want = oracle.decode_emit(host, v, n, kept_idx=kept).tobytes()
            assert outs[i].cpu().numpy().tobytes() == want, f"launch {i}"


@pytest.mark.parametrize("n,kept_frac", [(2504, None), (2504, 0.3), (40000, 0.01), (700, None)])
def test_hip_graph_capture_and_replay(n, kept_frac):
    """include/pgen_hip.h promises: no allocation, no synchronisation inside pgenhip_decode_emit, so a call
    can be captured into a HIP graph.  Capture one call (work-queue stream kernel, pick and scan-family kernels), replay it on three different record blocks, compare with the oracle."""
    rng = np.random.default_rng(31 + n)
    v = 257
    r = oracle.variant_record_size(n)
    kept = None if kept_frac is None else np.sort(rng.choice(n, size=int(n * kept_frac), replace=False)).astype(np.uint32)
    with pgen_rs_amd.GtEngine(n, kept_idx=kept, device=0) as eng:
        d_recs = torch.zeros(v * r, dtype=torch.uint8, device=DEV)
        out = torch.full((v * eng.gt_row_bytes,), SENTINEL, dtype=torch.uint8, device=DEV)
        side = torch.cuda.Stream(device=DEV)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            eng.use_torch_stream()
            eng.decode_emit(d_recs, v, out=out)  # warm-up outside the capture (module load, occupancy query)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            eng.use_torch_stream()                 # bind the ctx to the capturing stream
            eng.decode_emit(d_recs, v, out=out)
        for rep in range(3):
            recs = rng.integers(0, 256, size=v * r, dtype=np.uint8)
            d_recs.copy_(torch.from_numpy(recs))
            out.fill_(SENTINEL)
            g.replay()
            torch.cuda.synchronize()
            want = oracle.decode_emit(recs, v, n, kept_idx=kept)
            assert out.cpu().numpy().tobytes() == want.tobytes(), f"replay {rep}"
        eng.use_torch_stream()


def test_tune_rejects_unknown_knobs_and_values():
    with pgen_rs_amd.GtEngine(2504, device=0) as eng:
        for knob, value in ((99, 1), (_capi.KNOB_WIDE_RANGES, 3), (_capi.KNOB_WIDE_RANGES, 128), (_capi.KNOB_WIDE_RANGES, -2), (0, 0)):
            with pytest.raises(pgen_rs_amd.PgenHipError) as ei:
                eng.tune(knob, value)
            assert ei.value.status == _capi.ERR_BAD_ARG
        for knob in (_capi.KNOB_WIDE_BLOCKS_PER_CU, _capi.KNOB_WIDE_RANGES, _capi.KNOB_FLAT_BLOCKS_PER_CU, _capi.KNOB_SCAN_BLOCKS_PER_CU,
                     _capi.KNOB_PICK_BATCH_BYTES, _capi.KNOB_RUNS_ROWS, _capi.KNOB_SCAN_CHUNK_ROWS):
            eng.tune(knob, 0)   # 0 = back to the built-in default
        recs = eng.synth_records(50)
        out = eng.decode_emit(recs, 50)
        eng.wait()
        assert out.cpu().numpy().tobytes() == oracle.decode_emit(recs.cpu().numpy(), 50, 2504).tobytes()


def test_single_variant_and_zero_variants():
    n = 90
    recs = np.arange(oracle.variant_record_size(n), dtype=np.uint8)
    want = oracle.decode_emit(recs, 1, n).reshape(1, -1)
    got, _ = run_engine(recs, 1, n)
    assert (got == expect_buffer(want, 1, n, 4 * n + 1, 0, got.size)).all()
    got, _ = run_engine(recs, 0, n)
    assert (got == SENTINEL).all()


@pytest.mark.parametrize("n,kept_mod", [(19, None), (2504, None), (2504, 7), (5003, 100)])
def test_emit_lines_vs_oracle(n, kept_mod):
    rng = np.random.default_rng(400 + n)
    v = 23
    r = oracle.variant_record_size(n)
    recs = rng.integers(0, 256, size=v * r, dtype=np.uint8)
    kept = oracle.synth_keep(n, modulus=kept_mod) if kept_mod else None
    k = n if kept is None else kept.size
    prefixes = []
    for i in range(v):
        info = "x" * int(rng.integers(0, 200))
        prefixes.append(f"22\t{16050000 + 7 * i}\tsnp{i}\tA\tG\t100\tPASS\t{info}\tGT".encode())
    blob = np.frombuffer(b"".join(prefixes), dtype=np.uint8)
    poff = np.cumsum([0] + [len(p) for p in prefixes]).astype(np.int64)
    loff = np.cumsum([0] + [len(p) + 4 * k + 1 for p in prefixes]).astype(np.int64)
    want = oracle.emit_lines(recs, v, n, blob, poff.astype(np.uint64), loff.astype(np.uint64), kept_idx=kept)
    with pgen_rs_amd.GtEngine(n, kept_idx=kept, device=0) as eng:
        out = torch.full((int(loff[-1]) + 32,), SENTINEL, dtype=torch.uint8, device=DEV)
        eng.emit_lines(torch.from_numpy(recs).to(DEV), v, torch.from_numpy(blob.copy()).to(DEV), torch.from_numpy(poff).to(DEV),
                       torch.from_numpy(loff).to(DEV), max(len(p) for p in prefixes), out)
        eng.wait()
        got = out.cpu().numpy()
    assert bytes(got[: want.size]) == want.tobytes()
    assert (got[want.size :] == SENTINEL).all()


@pytest.mark.parametrize("n,v", [(61, 50), (300, 900), (1023, 120), (1024, 37), (2504, 700), (4099, 300), (70001, 40)])
@pytest.mark.parametrize("kernel", [_capi.KERNEL_ROWS, _capi.KERNEL_WIDE, _capi.KERNEL_PICK, _capi.KERNEL_AUTO])
def test_emit_lines_stream_kernel(n, v, kernel):
    """Full lines through the work-queue stream kernel (PGENHIP_KERNEL_WIDE) and the general kernel:
    prefixes of 0..40 bytes (so GT segments start at every byte phase and neighbouring lines share
    16-byte chunks), an unaligned output pointer, a gapped variant list, several spans per row
    (N = 70 001), many rows per block; sentinel bytes around the output must stay untouched."""
    if kernel == _capi.KERNEL_WIDE and n < 1024:
        pytest.skip("stream kernel: N >= 1024")
    if kernel == _capi.KERNEL_PICK and n > 4096:
        pytest.skip("pick kernel: short records only")
    rng = np.random.default_rng(500 + n)
    r = oracle.variant_record_size(n)
    v_file = v + 11
    recs = rng.integers(0, 256, size=v_file * r, dtype=np.uint8)
    vidx = np.sort(rng.choice(v_file, size=v, replace=False))
    prefixes = []
    for i in range(v):
        ln = int(rng.integers(0, 41)) if i % 7 else 0
        prefixes.append(bytes(rng.integers(33, 127, size=ln, dtype=np.uint8)))
    blob = np.frombuffer(b"".join(prefixes) + b"!", dtype=np.uint8)
    poff = np.cumsum([0] + [len(q) for q in prefixes]).astype(np.int64)
    loff = np.cumsum([0] + [len(q) + 4 * n + 1 for q in prefixes]).astype(np.int64)
    want = oracle.emit_lines(recs, v, n, blob, poff.astype(np.uint64), loff.astype(np.uint64), variant_idx=vidx)
    for out_offset in (0, 5):
        with pgen_rs_amd.GtEngine(n, device=0) as eng:
            out = torch.full((out_offset + int(loff[-1]) + 48,), SENTINEL, dtype=torch.uint8, device=DEV)
            eng.emit_lines(torch.from_numpy(recs).to(DEV), v, torch.from_numpy(blob.copy()).to(DEV), torch.from_numpy(poff).to(DEV),
                           torch.from_numpy(loff).to(DEV), 40, out[out_offset:],
                           variant_idx=torch.tensor(vidx, dtype=torch.int32, device=DEV), kernel=kernel)
            eng.wait()
            got = out.cpu().numpy()
        assert (got[:out_offset] == SENTINEL).all()
        body = got[out_offset : out_offset + want.size]
        if bytes(body) != want.tobytes():
            bad = np.flatnonzero(body != want)
            raise AssertionError(f"n={n} kernel={kernel} offset={out_offset}: {bad.size} bytes differ, first at {bad[:5]}")
        assert (got[out_offset + want.size :] == SENTINEL).all()


@pytest.mark.parametrize("n", [1024, 1025, 1399, 2504, 4099, 16415, 70001])
def test_emit_lines_long_rows_long_prefixes(n):
    """Full lines of rows >= 4 KiB through the stream kernel (GT segments in place behind their prefixes + the prefix copy):
    prefixes from 0 to 700 bytes (empty ones, one-byte ones behind the longest, the 130-250 bytes of the reference's own
    basic1.pvar rows), V = 1 and 2, a first line that does not start at output byte 0, a blob that does not start at a prefix,
    unaligned output and record pointers, a gapped variant list, several spans per row; sentinels around the output."""
    rng = np.random.default_rng(9700 + n)
    r = oracle.variant_record_size(n)
    for v, out_off, rec_off, lead_gap, pmax, gather in ((1, 0, 0, 0, 40, False), (2, 5, 3, 0, 480, False), (9, 15, 1, 7, 14, False), (41, 16, 15, 0, 260, True),
                                                          (130, 127, 7, 3, 480, False), (23, 1, 0, 0, 700, False), (300, 64, 0, 0, 200, True)):
        v = max(1, min(v, 3_000_000 // n))
        v_file = v + 5 if gather else v
        recs = rng.integers(0, 256, size=rec_off + v_file * r, dtype=np.uint8)
        vidx = np.sort(rng.choice(v_file, size=v, replace=False)) if gather else None
        plens = [int(rng.integers(0, pmax + 1)) if i % 5 else 0 for i in range(v)]
        if v > 3:
            plens[1], plens[2] = pmax, 1                                                  # the longest one, and a one-byte one behind it
        prefixes = [bytes(rng.integers(33, 127, size=q, dtype=np.uint8)) for q in plens]
        blob = np.frombuffer(b"?" * 3 + b"".join(prefixes) + b"!", dtype=np.uint8)
        poff = (3 + np.cumsum([0] + plens)).astype(np.int64)
        loff = (lead_gap + np.cumsum([0] + [q + 4 * n + 1 for q in plens])).astype(np.int64)
        want = oracle.emit_lines(recs[rec_off:], v, n, blob, poff.astype(np.uint64), (loff - lead_gap).astype(np.uint64), variant_idx=vidx)
        with pgen_rs_amd.GtEngine(n, device=0) as eng:
            eng.tune(_capi.KNOB_WIDE_BLOCKS_PER_CU, 1)
            out = torch.full((out_off + int(loff[-1]) + 48,), SENTINEL, dtype=torch.uint8, device=DEV)
            eng.emit_lines(torch.from_numpy(recs).to(DEV), v, torch.from_numpy(blob.copy()).to(DEV), torch.from_numpy(poff).to(DEV),
                           torch.from_numpy(loff).to(DEV), max(plens), out[out_off:], kernel=_capi.KERNEL_WIDE, records_offset=rec_off,
                           variant_idx=None if vidx is None else torch.tensor(vidx, dtype=torch.int32, device=DEV))
            eng.wait()
            got = out.cpu().numpy()
        body = got[out_off + lead_gap : out_off + lead_gap + want.size]
        if bytes(body) != want.tobytes():
            bad = np.flatnonzero(body != want)
            raise AssertionError(f"n={n} v={v} out_off={out_off} rec_off={rec_off} gap={lead_gap} pmax={pmax}: {bad.size} bytes differ, first at {bad[:8]}")
        assert (got[: out_off + lead_gap] == SENTINEL).all() and (got[out_off + lead_gap + want.size :] == SENTINEL).all(), f"n={n} v={v} wrote outside"


@pytest.mark.parametrize("keep", [None, 0.5, 0.1])
@pytest.mark.parametrize("n", [8, 9, 33, 61, 100, 300, 301, 302, 303, 500, 1000, 1023, 1500, 1900])
def test_emit_lines_runs_of_lines(n, keep):
    """Full lines on SHORT rows through the line-run kernel (a run of lines per work item, prefixes + GT text + '\n' assembled
    in LDS, whole-line stores): prefixes of 0..40 bytes (empty ones too, so GT segments start at every byte phase and seams fall
    everywhere in a chunk), V = 1, V smaller than a run, many runs per block, forced 2- and 3-line runs, unaligned output and
    record pointers, a first line that does not start at output byte 0; sentinel bytes around the output."""
    rng = np.random.default_rng(9100 + n)
    r = oracle.variant_record_size(n)
    kept = None
    if keep is not None:
        # kept-subset form: the picks go through the block's LDS copy of the kept list (>= 8 kept samples)
        if int(n * keep) < 8:
            pytest.skip("fewer than 8 kept samples")
        kept = np.sort(rng.choice(n, size=int(n * keep), replace=False)).astype(np.uint32)
    k = n if kept is None else int(kept.size)
    for v, out_off, rec_off, rows_knob, lead_gap, pmax in ((1, 0, 0, 0, 0, 40), (2, 5, 3, 0, 0, 40), (9, 15, 1, 2, 7, 25), (257, 16, 15, 3, 0, 40),
                                                             (3001, 127, 7, 0, 0, 12), (7001, 1, 0, 0, 3, 40)):
        recs = rng.integers(0, 256, size=rec_off + v * r, dtype=np.uint8)
        prefixes = [bytes(rng.integers(33, 127, size=int(rng.integers(0, pmax + 1)) if i % 5 else 0, dtype=np.uint8)) for i in range(v)]
        blob = np.frombuffer(b"?" * 3 + b"".join(prefixes) + b"!", dtype=np.uint8)      # the blob does not start at a prefix either
        poff = (3 + np.cumsum([0] + [len(q) for q in prefixes])).astype(np.int64)
        loff = (lead_gap + np.cumsum([0] + [len(q) + 4 * k + 1 for q in prefixes])).astype(np.int64)
        want = oracle.emit_lines(recs[rec_off:], v, n, blob, poff.astype(np.uint64), (loff - lead_gap).astype(np.uint64), kept_idx=kept)
        with pgen_rs_amd.GtEngine(n, kept_idx=kept, device=0) as eng:
            eng.tune(_capi.KNOB_RUNS_ROWS, rows_knob)
            if v > 5000:
                eng.tune(_capi.KNOB_WIDE_BLOCKS_PER_CU, 1)
            out = torch.full((out_off + int(loff[-1]) + 48,), SENTINEL, dtype=torch.uint8, device=DEV)
            try:
                eng.emit_lines(torch.from_numpy(recs).to(DEV), v, torch.from_numpy(blob.copy()).to(DEV), torch.from_numpy(poff).to(DEV),
                               torch.from_numpy(loff).to(DEV), pmax, out[out_off:], kernel=_capi.KERNEL_RUNS, records_offset=rec_off)
            except pgen_rs_amd.PgenHipError:
                # refusals that are in order: fewer than two lines per item (one wide load holds the run's records — and, with a kept
                # list, the whole record behind the run), or a line too long for half a span
                assert 1040 // r - (0 if kept is None else 1) < 2 or 4 * k + 1 + pmax > 7664, "line-run kernel refused a shape it should take"
                continue
            eng.wait()
            got = out.cpu().numpy()
        body = got[out_off + lead_gap : out_off + lead_gap + want.size]
        if bytes(body) != want.tobytes():
            bad = np.flatnonzero(body != want)
            raise AssertionError(f"n={n} k={k} v={v} out_off={out_off} rec_off={rec_off} rows={rows_knob} gap={lead_gap}: {bad.size} bytes differ, first at {bad[:8]}")
        assert (got[: out_off + lead_gap] == SENTINEL).all() and (got[out_off + lead_gap + want.size :] == SENTINEL).all()


@pytest.mark.parametrize("n,frac", [(2504, 0.01), (2504, 0.5), (40000, 0.01), (40000, 0.3), (40000, 0.9), (120000, 0.004), (120000, 0.02)])
@pytest.mark.parametrize("kernel", [_capi.KERNEL_AUTO, _capi.KERNEL_SCAN, _capi.KERNEL_ROWS, _capi.KERNEL_PICK])
def test_emit_lines_kept_subsets(n, frac, kernel):
    """Full lines with a sample filter: the subset kernel AUTO picks for the density and N (short-record pick,
    segment pick, list gather) writes its GT segments behind the prefixes and
    the prefix kernel fills those in; same bytes as the general kernel and the oracle, with a
    gapped variant list and sentinel bytes around the output."""
    if kernel == _capi.KERNEL_PICK and n > 4096:
        pytest.skip("pick kernel: short records only")
    rng = np.random.default_rng(int(700 + n + 1000 * frac))
    v = 150
    v_file = v + 9
    r = oracle.variant_record_size(n)
    recs = rng.integers(0, 256, size=v_file * r, dtype=np.uint8)
    vidx = np.sort(rng.choice(v_file, size=v, replace=False))
    kept = np.sort(rng.choice(n, size=max(4, int(n * frac)), replace=False)).astype(np.uint32)
    k = int(kept.size)
    prefixes = [bytes(rng.integers(33, 127, size=int(rng.integers(0, 60)) if i % 5 else 0, dtype=np.uint8)) for i in range(v)]
    blob = np.frombuffer(b"".join(prefixes) + b"!", dtype=np.uint8)
    poff = np.cumsum([0] + [len(q) for q in prefixes]).astype(np.int64)
    loff = np.cumsum([0] + [len(q) + 4 * k + 1 for q in prefixes]).astype(np.int64)
    want = oracle.emit_lines(recs, v, n, blob, poff.astype(np.uint64), loff.astype(np.uint64), kept_idx=kept, variant_idx=vidx)
    out_offset = 7
    with pgen_rs_amd.GtEngine(n, kept_idx=kept, device=0) as eng:
        out = torch.full((out_offset + int(loff[-1]) + 48,), SENTINEL, dtype=torch.uint8, device=DEV)
        eng.emit_lines(torch.from_numpy(recs).to(DEV), v, torch.from_numpy(blob.copy()).to(DEV), torch.from_numpy(poff).to(DEV),
                       torch.from_numpy(loff).to(DEV), 60, out[out_offset:],
                       variant_idx=torch.tensor(vidx, dtype=torch.int32, device=DEV), kernel=kernel)
        eng.wait()
        got = out.cpu().numpy()
    assert (got[:out_offset] == SENTINEL).all()
    body = got[out_offset : out_offset + want.size]
    if bytes(body) != want.tobytes():
        bad = np.flatnonzero(body != want)
        raise AssertionError(f"n={n} frac={frac} kernel={kernel}: {bad.size} bytes differ, first at {bad[:6]}")
    assert (got[out_offset + want.size :] == SENTINEL).all()


@pytest.mark.parametrize("n", [300, 2504, 70_001])
@pytest.mark.parametrize("k", [0, 1, 2, 3])
def test_emit_lines_tiny_keep_lists(n, k):
    """Full lines with K in {0, 1, 2, 3} (an EMPTY kept list must be "nobody", not "all samples" —
    each line is then prefix + '\n'); AUTO and every kernel that accepts the shape, with sentinel bytes around
    the output: a kernel that mistook the empty list for all samples would write 4N+1 bytes per line."""
    rng = np.random.default_rng(8800 + n + k)
    v = 211
    r = oracle.variant_record_size(n)
    recs = rng.integers(0, 256, size=v * r, dtype=np.uint8)
    kept = np.sort(rng.choice(n, size=k, replace=False)).astype(np.uint32)
    prefixes = [bytes(rng.integers(33, 127, size=int(rng.integers(0, 50)) if i % 4 else 0, dtype=np.uint8)) for i in range(v)]
    blob = np.frombuffer(b"".join(prefixes) + b"!", dtype=np.uint8)
    poff = np.cumsum([0] + [len(q) for q in prefixes]).astype(np.int64)
    loff = np.cumsum([0] + [len(q) + 4 * k + 1 for q in prefixes]).astype(np.int64)
    want = oracle.emit_lines(recs, v, n, blob, poff.astype(np.uint64), loff.astype(np.uint64), kept_idx=kept)
    assert want.size == sum(len(q) for q in prefixes) + v * (4 * k + 1)
    with pgen_rs_amd.GtEngine(n, kept_idx=kept, device=0) as eng:
        assert eng.kept_count == k and eng.gt_row_bytes == 4 * k + 1
        forced = (_capi.KERNEL_AUTO, _capi.KERNEL_ROWS, _capi.KERNEL_SCAN) + ((_capi.KERNEL_PICK,) if k >= 1 and n <= 4096 else ())
        for kernel in forced:
            out = torch.full((3 + int(loff[-1]) + 48,), SENTINEL, dtype=torch.uint8, device=DEV)
            eng.emit_lines(torch.from_numpy(recs).to(DEV), v, torch.from_numpy(blob.copy()).to(DEV), torch.from_numpy(poff).to(DEV),
                           torch.from_numpy(loff).to(DEV), 50, out[3:], kernel=kernel)
            eng.wait()
            got = out.cpu().numpy()
            assert (got[:3] == SENTINEL).all() and (got[3 + want.size :] == SENTINEL).all(), f"kernel {kernel} wrote outside the lines"
            assert bytes(got[3 : 3 + want.size]) == want.tobytes(), f"kernel {kernel}"
        # GT segments only (pgenhip_decode_emit): rows of 4K+1 bytes, K = 0 -> one '\n' per row
        for kernel in forced:
            out = torch.full((v * (4 * k + 1) + 32,), SENTINEL, dtype=torch.uint8, device=DEV)
            eng.decode_emit(torch.from_numpy(recs).to(DEV), v, out=out, kernel=kernel)
            eng.wait()
            got = out.cpu().numpy()
            assert bytes(got[: v * (4 * k + 1)]) == oracle.decode_emit(recs, v, n, kept_idx=kept).tobytes(), f"kernel {kernel}"
            assert (got[v * (4 * k + 1) :] == SENTINEL).all()


@pytest.mark.parametrize("hwe", [False, True])
def test_device_synth_matches_oracle_twin(hwe):
    for n, v, first, stride_pad, dirty in [(2504, 33, 0, 0, False), (10007, 9, 123, 3, not hwe), (5, 4, 2**31, 0, False), (500000, 2, 999_999, 0, False),
                                           (301, 700, 5, 1, False)]:
        r = oracle.variant_record_size(n)
        with pgen_rs_amd.GtEngine(n, device=0) as eng:
            t = eng.synth_records(v, first_variant=first, record_stride=r + stride_pad, dirty_pad=dirty, hwe=hwe)
            eng.wait()
            got = t.cpu().numpy()
        want = oracle.synth_records(n, v, first, record_stride=r + stride_pad, dirty_pad=dirty, hwe=hwe)
        got2 = got[: want.size].reshape(v, r + stride_pad)[:, :r]
        assert (got2 == want.reshape(v, r + stride_pad)[:, :r]).all()


def test_bad_arguments_are_status_codes_not_crashes():
    with pytest.raises(pgen_rs_amd.PgenHipError) as ei:
        pgen_rs_amd.GtEngine(10, kept_idx=[3, 10])
    assert ei.value.status == _capi.ERR_INDEX_RANGE
    with pytest.raises(pgen_rs_amd.PgenHipError) as ei:
        pgen_rs_amd.GtEngine(10, kept_idx=[3, 3])
    assert ei.value.status == _capi.ERR_BAD_ARG
    with pytest.raises(pgen_rs_amd.PgenHipError) as ei:
        pgen_rs_amd.GtEngine(10, device=99)
    assert ei.value.status == _capi.ERR_NO_DEVICE
    with pgen_rs_amd.GtEngine(10) as eng:
        recs = torch.zeros(30, dtype=torch.uint8, device=DEV)
        out = torch.zeros(1000, dtype=torch.uint8, device=DEV)
        with pytest.raises(pgen_rs_amd.PgenHipError) as ei:
            eng.decode_emit(recs, 3, out=out, out_stride=40)  # < 4K+1
        assert ei.value.status == _capi.ERR_BAD_ARG


def test_large_shape_properties_config3_rows():
    """BASELINE config 3 row geometry (N = 500 000) on a few rows: size-independent properties —
    every 4th byte is TAB, rows end in LF, slash column, and the allele columns re-encode to the
    input codes (decode -> re-encode round trip), plus equality with the oracle on those rows."""
    n, v, first = 500_000, 6, 34_357  # straddles the reference's u32 wrap index 34 360 (F5)
    with pgen_rs_amd.GtEngine(n, device=0) as eng:
        recs = eng.synth_records(v, first_variant=first)
        out = eng.decode_emit(recs, v)
        eng.wait()
        rows = out.cpu().numpy().reshape(v, 4 * n + 1)
        host = recs.cpu().numpy()
    assert (rows[:, -1] == ord("\n")).all()
    body = rows[:, :-1].reshape(v, n, 4)
    assert (body[:, :, 0] == 9).all() and (body[:, :, 2] == ord("/")).all()
    a, b = body[:, :, 1], body[:, :, 3]
    code = np.where(a == ord("."), 3, (a - ord("0")) + (b - ord("0"))).astype(np.uint8)
    shifts = np.array([0, 2, 4, 6], dtype=np.uint8)
    want_code = ((host.reshape(v, -1)[:, :, None] >> shifts) & 3).reshape(v, -1)[:, :n]
    assert (code == want_code).all()
    assert rows.tobytes() == oracle.decode_emit(host, v, n).tobytes()


@pytest.mark.parametrize("ranges,per_cu", [(1, 2), (2, 2), (4, 1), (8, 3), (2, 0), (0, 0), (16, 1), (64, 2)])
def test_wide_kernel_variants(ranges, per_cu):
    """The work-queue stream kernel (1 loader + 7 storer waves) against the oracle with every number of queue
    ranges and several grid sizes, on shapes with 1 and several spans per row."""
    tune = {_capi.KNOB_WIDE_RANGES: ranges, _capi.KNOB_WIDE_BLOCKS_PER_CU: per_cu}
    rng = np.random.default_rng(500 + ranges * 2 + per_cu)
    for n, v, off in [(2504, 301, 0), (1024, 77, 5), (4099, 40, 0), (40001, 9, 3), (70001, 5, 0)]:
        r = oracle.variant_record_size(n)
        recs = rng.integers(0, 256, size=v * r, dtype=np.uint8)
        want = oracle.decode_emit(recs, v, n).reshape(v, -1)
        got, _ = run_engine(recs, v, n, kernel=_capi.KERNEL_WIDE, out_offset=off, tune=tune)
        exp = expect_buffer(want, v, n, 4 * n + 1, off, got.size)
        assert (got == exp).all(), f"n={n} v={v} off={off}"
    vidx = [5, 0, 3, 3, 9]
    n = 3000
    recs = rng.integers(0, 256, size=10 * oracle.variant_record_size(n), dtype=np.uint8)
    want = oracle.decode_emit(recs, len(vidx), n, variant_idx=vidx).reshape(len(vidx), -1)
    got, _ = run_engine(recs, len(vidx), n, kernel=_capi.KERNEL_WIDE, variant_idx=vidx, tune=tune)
    assert (got == expect_buffer(want, len(vidx), n, 4 * n + 1, 0, got.size)).all()


def test_wide_kernel_every_phase_of_the_row_seam():
    """Row items of the stream kernel: the seam between two rows (last chunk of row j with its '\\n' and the head of row j+1, first
    store step of row j+1 starting mid-KiB) at every phase.  N = 1 024 (S = 4 097: the row start moves one byte per row, so 301
    rows see every phase of a 128-B line and of a chunk), rows whose '\\n' is the last byte of a chunk, several spans per row,
    every alignment class of the output pointer, gathered and padded records (the first byte of row j+1 comes from wherever
    that row's record lies), V = 1 and 2."""
    rng = np.random.default_rng(8100)
    tune = {_capi.KNOB_WIDE_BLOCKS_PER_CU: 1}
    for n, v, off in [(1024, 301, 0), (1024, 140, 127), (1055, 97, 1), (2504, 260, 16), (2504, 2, 113), (2527, 33, 15), (4099, 1, 64),
                      (16415, 21, 0), (40001, 9, 3), (70001, 5, 120)]:
        r = oracle.variant_record_size(n)
        recs = rng.integers(0, 256, size=v * r, dtype=np.uint8)
        want = oracle.decode_emit(recs, v, n).reshape(v, -1)
        got, _ = run_engine(recs, v, n, kernel=_capi.KERNEL_WIDE, out_offset=off, tune=tune)
        exp = expect_buffer(want, v, n, 4 * n + 1, off, got.size)
        if not (got == exp).all():
            bad = np.flatnonzero(got != exp)
            raise AssertionError(f"n={n} v={v} off={off}: {bad.size} bytes differ, first at {bad[:8]}")
    # gathered rows (variant list with repeats and back-steps) and padded records at an odd base address
    n = 3000
    r = oracle.variant_record_size(n)
    for rstride, rec_off in ((r, 0), (r + 5, 3)):
        v_file = 40
        recs = rng.integers(0, 256, size=rec_off + v_file * rstride, dtype=np.uint8)
        vidx = np.array([5, 0, 3, 3, 39, 38, 1, 2, 2, 17] * 7, dtype=np.uint32)
        dense = np.concatenate([recs[rec_off + i * rstride : rec_off + i * rstride + r] for i in range(v_file)])
        want = oracle.decode_emit(dense, len(vidx), n, variant_idx=vidx).reshape(len(vidx), -1)
        for off in (0, 77):
            got, _ = run_engine(recs, len(vidx), n, kernel=_capi.KERNEL_WIDE, variant_idx=vidx, record_stride=rstride, records_offset=rec_off,
                                out_offset=off, tune=tune)
            assert (got == expect_buffer(want, len(vidx), n, 4 * n + 1, off, got.size)).all(), f"gathered stride={rstride} off={off}"


@pytest.mark.parametrize("v", [100_000, 125_000])
def test_config3_full_size_100k_by_500k(v):
    """BASELINE config 3 at its full size in ONE launch: 100 000 variants x 500 000 samples,
    12.5 GB of records -> 200 GB of text (64-bit offsets everywhere); and the 125 000-variant shard one
    of 8 GPUs owns in config 4 (15.6 GB -> 250 GB, 266 GB resident).  Checked through
    size-independent properties on the whole buffer (every row ends in LF at the right place,
    TAB/slash columns on a strided sample) and byte equality with the oracle on rows picked from
    the start, the u32-wrap boundary of the reference (34 359/34 360), the middle and the end."""
    n = 500_000
    torch.cuda.empty_cache()
    free, _total = torch.cuda.mem_get_info(0)
    need = v * (125_000 + 4 * n + 1) + (2 << 30)
    if free < need:
        pytest.skip(f"needs {need / 2**30:.0f} GiB of free HBM, have {free / 2**30:.0f}")
    row = 4 * n + 1
    with pgen_rs_amd.GtEngine(n, device=0) as eng:
        recs = eng.synth_records(v)
        out = eng.decode_emit(recs, v)
        eng.wait()
        assert out.numel() == v * row
        # every row's last byte is LF: strided view over the whole 200 GB
        lf = out[row - 1 :: row]
        assert lf.numel() == v and bool((lf == 10).all())
        # first byte of every row is TAB
        assert bool((out[0::row] == 9).all())
        for j in (0, 1, 34_359, 34_360, 50_000, 99_998, v - 2, v - 1):
            got = out[j * row : (j + 1) * row].cpu().numpy()
            host = recs[j * 125_000 : (j + 1) * 125_000].cpu().numpy()
            assert got.tobytes() == oracle.decode_emit(host, 1, n).tobytes(), f"row {j}"
            assert host.tobytes() == oracle.synth_records(n, 1, first_variant=j).tobytes()
        del out, recs
    torch.cuda.empty_cache()


@pytest.mark.parametrize("v", [100_000, 125_000])
@pytest.mark.parametrize("path", ["two_pass", "two_pass_segment_compact", "row_owner_single_pass", "segment_xcd", "segment_plain"])
def test_config5_geometry_500k_samples_keep_1pct(path, v):
    """BASELINE config 5's per-GPU geometry in ONE launch: 100 000 variants x 500 000 samples and the
    125 000-variant shard each of 8 GPUs owns (12.5 / 15.6 GB of records, offsets beyond 2^32), the 1 %
    splitmix keep mask of SURVEY 8(d) (4 940 kept -> 19 761-byte rows, 1.98 / 2.47 GB of text).
    AUTO (two passes: compact records — from the row-owner kernel, or from the segment kernel — then the row-item stream
    kernel on them), the row-owner kernel writing text in one pass, and the single-pass segment kernel with the XCD-aware and
    the plain block map: LF / TAB / slash columns over the whole buffer, byte
    equality with the oracle on rows from the start, the reference's u32-wrap boundary, the middle and
    the end, and equality of the three paths' whole outputs through a checksum of checksums."""
    n = 500_000
    free, _total = torch.cuda.mem_get_info(0)
    if free < v * 125_000 + (8 << 30):
        pytest.skip("needs ~21 GiB of free HBM")
    kept = oracle.synth_keep(n, modulus=100)
    k = int(kept.size)
    row = 4 * k + 1
    with pgen_rs_amd.GtEngine(n, kept_idx=kept, device=0) as eng:
        eng.tune(_capi.KNOB_SCAN_XCD_MAP, -1 if path == "segment_plain" else 1)
        eng.tune(_capi.KNOB_SCAN_ROWPICK, {"two_pass_segment_compact": -1, "row_owner_single_pass": 2}.get(path, 1))
        recs = eng.synth_records(v)
        out = eng.decode_emit(recs, v, kernel=_capi.KERNEL_SCAN if path.startswith("segment") else _capi.KERNEL_AUTO)
        eng.wait()
        assert out.numel() == v * row
        assert bool((out[row - 1 :: row] == 10).all())
        body = out.view(v, row)[:, :-1].reshape(v, k, 4)
        assert bool((body[:, :, 0] == 9).all()) and bool((body[:, :, 2] == 47).all())
        for j in (0, 1, 34_359, 34_360, 34_361, 50_000, 77_777, 99_998, v - 2, v - 1):
            got = out[j * row : (j + 1) * row].cpu().numpy()
            host = oracle.synth_records(n, 1, first_variant=j)
            assert got.tobytes() == oracle.decode_emit(host, 1, n, kept_idx=kept).tobytes(), f"row {j}"
        # checksum of per-row checksums (int64 sums of the bytes weighted by column): equal for both kernels
        w = torch.arange(1, row + 1, dtype=torch.int64, device=DEV)
        sums = (out.view(v, row).to(torch.int64) * w).sum(dim=1)
        digest = int((sums * torch.arange(1, v + 1, dtype=torch.int64, device=DEV)).sum().item())
        del body, out, recs, sums
    torch.cuda.empty_cache()
    seen = _CONFIG5_DIGEST.setdefault(v, digest)
    assert seen == digest, "the subset kernels disagree somewhere in the 1.98 GB of text"


_CONFIG5_DIGEST = {}


@pytest.mark.parametrize("ranges", [1, 2, 8, 32])
def test_wide_kernel_many_steps_ring_reuse(ranges):
    """Enough items per block that the loader/storer LDS ring is reused many times and the work queue
    is drained and stolen from (one block per CU, 20 000 rows), whole output compared with the oracle."""
    n, v = 2504, 20_000
    with pgen_rs_amd.GtEngine(n, device=0) as eng:
        eng.tune(_capi.KNOB_WIDE_BLOCKS_PER_CU, 1)
        eng.tune(_capi.KNOB_WIDE_RANGES, ranges)
        recs = eng.synth_records(v, first_variant=77)
        out = eng.decode_emit(recs, v, kernel=_capi.KERNEL_WIDE)
        eng.wait()
        got = out.cpu().numpy()
        host = recs.cpu().numpy()
    assert got.tobytes() == oracle.decode_emit(host, v, n).tobytes()


def _random_case(rng):
    """One random call of the path: shape, keep list, gather, strides, pointer phases, GT segments or full lines."""
    # sample counts clustered around the dispatch thresholds of capi.hip (8, 61, 300, 400, 768, 1000, 1024, 1400, 1916, 4096, 65536) and spread between
    edges = [1, 2, 7, 8, 9, 60, 61, 64, 299, 300, 301, 399, 400, 401, 767, 768, 769, 999, 1000, 1001, 1023, 1024, 1025, 1399, 1400, 1401, 1915, 1916, 1917, 2504, 4095, 4096, 4097, 20011,
             65535, 65536, 65537]
    n = int(rng.choice(edges)) if rng.random() < 0.6 else int(rng.integers(1, 30_000))
    budget = int(rng.choice([300_000, 1_500_000, 6_000_000]))   # genotypes scanned per case: keeps the oracle in milliseconds
    v = int(min(max(1, budget // n), rng.choice([1, 2, 3, 17, 64, 257, 1031, 5003, 20011])))
    mode = rng.choice(["all", "all", "dense", "sparse", "tiny", "modulus"])
    if mode == "all" or n < 2:
        kept = None
    elif mode == "dense":
        kept = np.sort(rng.choice(n, size=max(1, int(n * rng.uniform(0.2, 0.98))), replace=False))
    elif mode == "sparse":
        kept = np.sort(rng.choice(n, size=max(1, int(n * rng.uniform(0.002, 0.06))), replace=False))
    elif mode == "tiny":
        kept = np.sort(rng.choice(n, size=int(rng.integers(1, min(n, 5) + 1)), replace=False))
    else:
        kept = np.arange(int(rng.integers(0, 7)), n, int(rng.integers(2, 130)))
    lines = bool(rng.random() < 0.4)
    gather = bool(rng.random() < 0.3)
    pad = int(rng.choice([0, 0, 0, 1, 5, 16])) if not lines or gather else 0
    return dict(n=n, v=v, kept=None if kept is None or kept.size == 0 else kept.astype(np.uint32), lines=lines, gather=gather, rec_pad=pad,
                out_off=int(rng.integers(0, 130)), rec_off=int(rng.integers(0, 18)), out_pad=int(rng.choice([0, 0, 0, 3, 16])) if not lines else 0)


@pytest.mark.parametrize("seed", range(40))
def test_randomized_differential_auto_dispatch(seed):
    """Seeded random calls through the AUTO dispatch (whatever kernel capi.hip picks for the shape) against the oracle: random N
    (clustered at every dispatch threshold), V, keep lists (none / dense / sparse / tiny / strided), variant gathers with repeats,
    padded record and output strides, unaligned record and output pointers, GT segments and full lines with random prefixes.
    Sentinel bytes everywhere the call must not write.  40 seeds x 40 cases."""
    rng = np.random.default_rng(77_000 + seed)
    for case_i in range(40):
        c = _random_case(rng)
        n, v, kept = c["n"], c["v"], c["kept"]
        r = oracle.variant_record_size(n)
        k = n if kept is None else int(kept.size)
        rstride = r + c["rec_pad"]
        v_file = v * 2 if c["gather"] else v
        recs = rng.integers(0, 256, size=c["rec_off"] + v_file * rstride + 16, dtype=np.uint8)
        vidx = rng.integers(0, v_file, size=v).astype(np.uint32) if c["gather"] else None
        dense = np.concatenate([recs[c["rec_off"] + i * rstride : c["rec_off"] + i * rstride + r] for i in range(v_file)]) if r else np.zeros(0, np.uint8)
        tag = f"seed={seed} case={case_i} {dict((q, (w if not isinstance(w, np.ndarray) else f'K={w.size}')) for q, w in c.items())}"
        with pgen_rs_amd.GtEngine(n, kept_idx=kept, device=0) as eng:
            rec_t = torch.from_numpy(recs).to(DEV)
            vidx_t = None if vidx is None else torch.tensor(vidx.astype(np.int64), dtype=torch.int32, device=DEV)
            if c["lines"]:
                plens = rng.integers(0, 60, size=v)
                plens[rng.random(v) < 0.2] = 0
                prefixes = [bytes(rng.integers(33, 127, size=int(q), dtype=np.uint8)) for q in plens]
                blob = np.frombuffer(b"".join(prefixes) + b"!", dtype=np.uint8)
                poff = np.cumsum([0] + [len(q) for q in prefixes]).astype(np.int64)
                loff = np.cumsum([0] + [len(q) + 4 * k + 1 for q in prefixes]).astype(np.int64)
                want = oracle.emit_lines(dense, v, n, blob, poff.astype(np.uint64), loff.astype(np.uint64), kept_idx=kept, variant_idx=vidx)
                out = torch.full((c["out_off"] + int(loff[-1]) + 64,), SENTINEL, dtype=torch.uint8, device=DEV)
                eng.emit_lines(rec_t, v, torch.from_numpy(blob.copy()).to(DEV), torch.from_numpy(poff).to(DEV), torch.from_numpy(loff).to(DEV),
                               int(max(plens.max(), 1)), out[c["out_off"]:], record_stride=rstride, variant_idx=vidx_t, records_offset=c["rec_off"])
                eng.wait()
                got = out.cpu().numpy()
                exp = np.full(got.size, SENTINEL, dtype=np.uint8)
                exp[c["out_off"] : c["out_off"] + want.size] = want
            else:
                ostride = 4 * k + 1 + c["out_pad"]
                want = oracle.decode_emit(dense, v, n, kept_idx=kept, variant_idx=vidx).reshape(v, -1)
                out = torch.full((c["out_off"] + v * ostride + 64,), SENTINEL, dtype=torch.uint8, device=DEV)
                eng.decode_emit(rec_t, v, record_stride=rstride, variant_idx=vidx_t, out=out, out_stride=ostride, out_offset=c["out_off"],
                                records_offset=c["rec_off"])
                eng.wait()
                got = out.cpu().numpy()
                exp = expect_buffer(want, v, k, ostride, c["out_off"], got.size)
        if not (got == exp).all():
            bad = np.flatnonzero(got != exp)
            raise AssertionError(f"{tag}: {bad.size} bytes differ, first at {bad[:8]}")


@pytest.mark.parametrize("n,v,keep_mod", [(300, 4_300_000, 0), (2504, 500_000, 0), (900, 1_350_000, 0), (40_000, 900_000, 29), (100_000, 26_000, 2), (60_000, 235_000, 12)])
def test_emit_lines_past_4_gib(n, v, keep_mod):
    """Full lines whose offsets pass 2^32 in ONE call (5.3 GB of text at the reference's own 300-sample shape through the line-run
    kernel, 5.0 GB at N = 2 504 through the stream kernel, 4.9 GB at N = 900 through the pick kernel, 5 GB through the two-pass
    path, 5.2 GB through the segment kernel's four-pick flush at half of 100 000 samples kept, 4.4 GB through the row-owner kernel at 8 % of 60 000): every kernel carries run- / row-relative offsets in 32 bits and the place of the run in 64.  Checked against the oracle
    on windows of lines at the start, around the 4-GiB mark, at the end and at random places; LF at the end of every line
    and sentinels behind the last one."""
    free, _ = torch.cuda.mem_get_info()
    kept = oracle.synth_keep(n, modulus=keep_mod) if keep_mod else None
    k = n if kept is None else int(kept.size)
    rng = np.random.default_rng(4400 + n)
    plen = rng.integers(0, 41, size=v).astype(np.int64)
    plen[::7] = 0
    poff = np.concatenate([[0], np.cumsum(plen)]).astype(np.int64)
    loff = np.concatenate([[0], np.cumsum(plen + 4 * k + 1)]).astype(np.int64)
    total = int(loff[-1])
    assert total > (1 << 32) + (1 << 28)
    if free < total + (3 << 30):
        pytest.skip("needs the output resident")
    blob = rng.integers(33, 127, size=int(poff[-1]) + 1, dtype=np.uint8)
    with pgen_rs_amd.GtEngine(n, kept_idx=kept, device=0) as eng:
        recs = eng.synth_records(v)
        out = torch.full((total + 64,), SENTINEL, dtype=torch.uint8, device=DEV)
        eng.emit_lines(recs, v, torch.from_numpy(blob).to(DEV), torch.from_numpy(poff).to(DEV), torch.from_numpy(loff).to(DEV), 40, out)
        eng.wait()
        r = eng.record_size
        assert (out[total:] == SENTINEL).all().item()
        ends = torch.from_numpy(loff[1:] - 1).to(DEV)
        assert (out[ends] == 10).all().item(), "a line does not end in LF"
        j_4g = int(np.searchsorted(loff, 1 << 32))
        starts = [0, max(j_4g - 40, 0), v - 60] + [int(q) for q in rng.integers(0, v - 60, size=6)]
        for j0 in starts:
            j1 = min(j0 + 60, v)
            host_recs = recs[j0 * r : j1 * r].cpu().numpy()
            want = oracle.emit_lines(host_recs, j1 - j0, n, blob, poff[j0 : j1 + 1].astype(np.uint64), (loff[j0 : j1 + 1] - loff[j0]).astype(np.uint64), kept_idx=kept)
            got = out[int(loff[j0]) : int(loff[j1])].cpu().numpy()
            if bytes(got) != want.tobytes():
                bad = np.flatnonzero(got != want)
                raise AssertionError(f"n={n} lines {j0}..{j1}: {bad.size} bytes differ, first at {bad[:6]}")
        del out


@pytest.mark.parametrize("n,v,keep_frac,pad", [(300, 4_300_000, 0.0, 0), (1000, 1_250_000, 0.0, 0), (2504, 900_000, 0.5, 0), (2504, 3_000_000, 0.15, 0),
                                                   (300, 4_000_000, 0.0, 3), (12, 95_000_000, 0.0, 0)])
def test_gt_segments_past_4_gib(n, v, keep_frac, pad):
    """GT segments whose byte offsets pass 2^32 in ONE call on SHORT records (the long-record kernels have the 200-GB tests): RUNS mode
    at the reference's own 300-sample shape (5.2 GB) and at N = 1 000, the pick kernel with half and 15 % of 2 504 samples kept,
    the flat kernel on padded rows, tiny rows in the RUNS mode (N = 12, 95 M rows).  Windows of rows at the start, around
    the 4-GiB mark, at the end and at random places against the oracle; LF at the end of every row; sentinels in the padding
    and behind the last row."""
    free, _ = torch.cuda.mem_get_info()
    rng = np.random.default_rng(4500 + n + v % 1000)
    kept = np.sort(rng.choice(n, size=int(n * keep_frac), replace=False)).astype(np.uint32) if keep_frac else None
    k = n if kept is None else int(kept.size)
    row = 4 * k + 1
    ostride = row + pad
    total = v * ostride
    assert total > (1 << 32) + (1 << 27)
    if free < total + (3 << 30):
        pytest.skip("needs the output resident")
    with pgen_rs_amd.GtEngine(n, kept_idx=kept, device=0) as eng:
        recs = eng.synth_records(v)
        out = torch.full((total + 64,), SENTINEL, dtype=torch.uint8, device=DEV)
        eng.decode_emit(recs, v, out=out, out_stride=ostride)
        eng.wait()
        r = eng.record_size
        assert (out[total - pad :] == SENTINEL).all().item()
        rows2d = out[:total].view(v, ostride)
        assert (rows2d[:, row - 1] == 10).all().item(), "a row does not end in LF"
        if pad:
            assert (rows2d[:, row:] == SENTINEL).all().item(), "padding bytes written"
        j_4g = (1 << 32) // ostride
        starts = [0, max(j_4g - 50, 0), v - 100] + [int(q) for q in rng.integers(0, v - 100, size=6)]
        for j0 in starts:
            j1 = min(j0 + 100, v)
            want = oracle.decode_emit(recs[j0 * r : j1 * r].cpu().numpy(), j1 - j0, n, kept_idx=kept).reshape(j1 - j0, row)
            got = rows2d[j0:j1, :row].cpu().numpy()
            if not (got == want).all():
                bad = np.argwhere(got != want)
                raise AssertionError(f"n={n} k={k} rows {j0}..{j1}: {bad.shape[0]} bytes differ, first at row/col {bad[:4].tolist()}")
        del out, rows2d


@pytest.mark.parametrize("four", [1, -1])
@pytest.mark.parametrize("unroll", [1, 2, 4])
@pytest.mark.parametrize("kernel", ["segment", "row_owner"])
def test_text_flush_four_picks_and_unrolls(kernel, four, unroll):
    """The subset kernels' text flush — four picks per 16-byte chunk with the fifth text from the next lane (wave_shl DPP, lane 63 from the
    next group / the chunk behind the step), store instructions aligned to 128-byte lines; round 2's five picks — with 1 / 2 / 4 chunks per lane
    and step (4: the segment kernel only): same bytes as the oracle for ragged record tails, segments with 0 / 1 / all samples kept, every row alignment and phase
    (odd K: rows start at every byte offset mod 16), with and without a gathered variant list, GT segments and full lines."""
    rng = np.random.default_rng(1000 + 10 * (four + 1) + unroll)
    kern = _capi.KERNEL_SCAN if kernel == "segment" else _capi.KERNEL_ROWPICK
    shapes = ((16385, 0.5, 37), (40_001, 0.93, 23), (70_003, 0.07, 29), (33_000, 1.0, 11)) if kernel == "segment" else ((16385, 0.5, 37), (70_003, 0.07, 29), (200_001, 0.08, 9))
    for n, dens, v in shapes:
        r = oracle.variant_record_size(n)
        keep = rng.random(n) < dens
        keep[16384:16384 + 900] = False          # a stretch without kept samples across a segment boundary
        if n > 33_000:
            keep[32768:32768 + 16384] = n % 2 == 0 and kernel == "segment"    # a segment that is empty (or wholly kept)
        kept = np.flatnonzero(keep).astype(np.uint32)
        if kept.size == n:
            kept = kept[:-1]
        recs = rng.integers(0, 256, size=2 * v * r, dtype=np.uint8)
        vidx = rng.permutation(2 * v)[:v]
        tune = {_capi.KNOB_SCAN_FOUR_PICKS: four, _capi.KNOB_FLUSH_UNROLL: unroll, _capi.KNOB_SCAN_BLOCKS_PER_CU: 1, _capi.KNOB_ROWPICK_BLOCKS_PER_CU: 1}
        got, k = run_engine(recs, v, n, kept=kept, kernel=kern, tune=tune)
        want = oracle.decode_emit(recs[: v * r], v, n, kept_idx=kept)
        assert bytes(got[: want.size]) == want.tobytes(), (n, dens)
        assert (got[want.size :] == SENTINEL).all()
        got, k = run_engine(recs, v, n, kept=kept, kernel=kern, tune=tune, variant_idx=vidx)
        want = oracle.decode_emit(recs.reshape(2 * v, r)[vidx].reshape(-1), v, n, kept_idx=kept)
        assert bytes(got[: want.size]) == want.tobytes(), (n, dens, "gathered")
        # full lines with prefixes of every length mod 16 (every phase of the GT text against the 16-byte chunks and the 128-byte lines)
        plen = ((np.arange(v) * 7 + 3) % 41 + 2).astype(np.int64)
        poff = np.concatenate([[0], np.cumsum(plen)]).astype(np.int64)
        loff = np.concatenate([[0], np.cumsum(plen + 4 * k + 1)]).astype(np.int64)
        blob_np = rng.integers(65, 91, size=int(poff[-1]), dtype=np.uint8)
        want_lines = oracle.emit_lines(recs[: v * r], v, n, blob_np, poff.astype(np.uint64), loff.astype(np.uint64), kept_idx=kept).tobytes()
        with pgen_rs_amd.GtEngine(n, kept_idx=kept, device=0) as eng:
            for knob, value in tune.items():
                eng.tune(knob, value)
            out = torch.full((int(loff[-1]) + 64,), SENTINEL, dtype=torch.uint8, device=DEV)
            eng.emit_lines(torch.from_numpy(recs[: v * r].copy()).to(DEV), v, torch.from_numpy(blob_np.copy()).to(DEV), torch.from_numpy(poff).to(DEV),
                           torch.from_numpy(loff).to(DEV), int(plen.max()), out, kernel=kern)
            eng.wait()
            got_l = out.cpu().numpy()
        assert bytes(got_l[: int(loff[-1])]) == want_lines, (n, dens, "lines")
        assert (got_l[int(loff[-1]) :] == SENTINEL).all()


@pytest.mark.parametrize("n,v,keep_mod", [(100_000, 26_000, 2), (60_000, 235_000, 12), (2504, 950_000, 2)])
def test_gt_segments_of_kept_subsets_past_4_gib(n, v, keep_mod):
    """GT segments of kept subsets whose output passes 2^32 bytes in one call: the segment kernel (half of 100 000 samples), the
    row-owner kernel (8 % of 60 000) and the short-record pick kernel (half of 2 504) — row- / piece-relative offsets in 32 bits, the
    row's place in 64.  LF at the end of every row over the whole buffer, sentinels behind it, byte equality with the oracle on rows
    at the start, around the 4-GiB mark and at the end."""
    kept = oracle.synth_keep(n, modulus=keep_mod)
    k = int(kept.size)
    row = 4 * k + 1
    total = v * row
    assert total > (1 << 32) + (1 << 27)
    free, _ = torch.cuda.mem_get_info()
    if free < total + v * oracle.variant_record_size(n) + (3 << 30):
        pytest.skip("needs records and output resident")
    with pgen_rs_amd.GtEngine(n, kept_idx=kept, device=0) as eng:
        recs = eng.synth_records(v)
        out = torch.full((total + 64,), SENTINEL, dtype=torch.uint8, device=DEV)
        eng.decode_emit(recs, v, out=out)
        eng.wait()
        assert (out[total:] == SENTINEL).all().item()
        assert (out[row - 1 : total : row] == 10).all().item()
        assert (out[0:total:row] == 9).all().item()
        j_4g = (1 << 32) // row
        for j in (0, 1, j_4g - 1, j_4g, j_4g + 1, v // 2, v - 2, v - 1):
            got = out[j * row : (j + 1) * row].cpu().numpy()
            host = oracle.synth_records(n, 1, first_variant=j)
            assert got.tobytes() == oracle.decode_emit(host, 1, n, kept_idx=kept).tobytes(), f"row {j}"
