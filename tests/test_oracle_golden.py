"""CPU leg: the C oracle against the committed golden vectors (independent numpy decoder),
the hand truth table, the header/offset semantics of the reference, and its own file-based
restatement.  No GPU, no product code."""
import hashlib

import numpy as np
import pytest

import pgen_oracle as oracle
from helpers import case_names, load_case, sha_table


def test_truth_table_e4():
    # record byte 0b11_10_01_00 -> 0/0 0/1 1/1 ./.   (src/pfile.rs:172-183)
    out = oracle.decode_emit(np.array([0xE4], dtype=np.uint8), 1, 4)
    assert bytes(out) == b"\t0/0\t0/1\t1/1\t./.\n"


@pytest.mark.parametrize("name", case_names())
def test_oracle_matches_golden_case(name):
    v, n, recs, kept, gt = load_case(name)
    out = oracle.decode_emit(recs.reshape(-1), v, n, kept_idx=kept)
    assert out.size == gt.size
    assert bytes(out) == bytes(gt)


@pytest.mark.parametrize("name", sorted(sha_table().keys()))
def test_oracle_matches_golden_sha(name):
    spec = sha_table()[name]
    n, v = spec["sample_count"], spec["n_variants"]
    recs = oracle.synth_records(n, v, spec["first_variant"], spec["seed"], dirty_pad=spec["dirty_pad"], hwe=spec["distribution"] == "hwe")
    assert hashlib.sha256(recs.tobytes()).hexdigest() == spec["records_sha256"]  # C generator == numpy generator
    kept = None
    if spec["keep_modulus"]:
        kept = oracle.synth_keep(n, spec["keep_seed"], spec["keep_modulus"])
        assert kept.size == spec["kept_count"]
    out = oracle.decode_emit(recs, v, n, kept_idx=kept)
    assert out.size == spec["gt_bytes"]
    assert hashlib.sha256(out.tobytes()).hexdigest() == spec["gt_sha256"]


def test_record_size_matches_reference_formula():
    # src/pfile.rs:196-200
    assert oracle.variant_record_size(2504) == 626
    assert oracle.variant_record_size(500000) == 125000
    assert oracle.variant_record_size(300) == 75
    for n in range(0, 70):
        assert oracle.variant_record_size(n) == (2 * n + 7) // 8


def test_header_asserts():
    good = bytes([0x6C, 0x1B, 0x02]) + (17784).to_bytes(4, "little") + (2504).to_bytes(4, "little") + b"\x40"
    assert oracle.parse_header(good) == (0, 17784, 2504)
    assert oracle.parse_header(b"\x6c\x1c" + good[2:])[0] == -1  # :47
    assert oracle.parse_header(good[:2] + b"\x10" + good[3:])[0] == -2  # :53
    assert oracle.parse_header(good[:11] + b"\x00")[0] == -3  # :69


def test_reference_u32_offset_wrap_is_documented():
    # src/pfile.rs:165 multiplies in u32.  At N=500k (R=125000) the last
    # correct variant index is 34359; 34360 wraps.  The build matches the exact offset.
    r = 125000
    assert oracle.record_offset_exact(34359, r) == oracle.record_offset_ref_u32_wrap(34359, r) == 12 + 34359 * r
    assert oracle.record_offset_exact(34360, r) == 12 + 34360 * r
    assert oracle.record_offset_ref_u32_wrap(34360, r) == 12 + (34360 * r) % 2**32
    assert oracle.record_offset_ref_u32_wrap(34360, r) != oracle.record_offset_exact(34360, r)
    # at chr22/basic1 geometry (R=626) nothing wraps below 6.86M variants
    assert oracle.record_offset_ref_u32_wrap(6_860_969, 626) == oracle.record_offset_exact(6_860_969, 626)


def test_strides_and_variant_gather():
    rng = np.random.default_rng(1)
    n, v = 37, 9
    r = oracle.variant_record_size(n)
    stride = r + 5
    recs = rng.integers(0, 256, size=v * stride, dtype=np.uint8)
    dense = np.concatenate([recs[i * stride : i * stride + r] for i in range(v)])
    want = oracle.decode_emit(dense, v, n).reshape(v, -1)
    got = oracle.decode_emit(recs, v, n, record_stride=stride, out_stride=4 * n + 1 + 3).reshape(v, -1)
    assert (got[:, : 4 * n + 1] == want).all() and (got[:, 4 * n + 1 :] == 0).all()
    vidx = [8, 0, 3, 3]
    got = oracle.decode_emit(recs, len(vidx), n, record_stride=stride, variant_idx=vidx).reshape(len(vidx), -1)
    assert (got == want[vidx]).all()


def test_kept_index_out_of_range_is_an_error():
    with pytest.raises(IndexError):
        oracle.decode_emit(np.zeros(2, dtype=np.uint8), 1, 5, kept_idx=[5])


def test_file_restatement_equals_memory_restatement(tmp_path):
    # the literal seek/read/BufWriter loop (src/pfile.rs:149-192) and the in-memory loop agree
    rng = np.random.default_rng(2)
    n, v = 2504, 40
    r = oracle.variant_record_size(n)
    recs = rng.integers(0, 256, size=v * r, dtype=np.uint8)
    pgen = tmp_path / "t.pgen"
    pgen.write_bytes(bytes([0x6C, 0x1B, 0x02]) + v.to_bytes(4, "little") + n.to_bytes(4, "little") + b"\x40" + recs.tobytes())
    vidx = [0, 5, 6, 39]
    kept = np.arange(1, n, 3, dtype=np.uint32)
    out = tmp_path / "o.vcf"
    assert oracle.output_vcf_body_file(str(pgen), n, str(out), var_idx=vidx, kept_idx=kept) == 0
    want = oracle.decode_emit(recs, len(vidx), n, kept_idx=kept, variant_idx=vidx)
    assert out.read_bytes() == want.tobytes()
    prefixes = [f"22\t{100 + i}\tsnp{i}\tA\tG\t.\t.\t.\tGT".encode() for i in vidx]
    assert oracle.output_vcf_body_file(str(pgen), n, str(out), var_idx=vidx, prefixes=prefixes) == 0
    body = out.read_bytes()
    seg = oracle.decode_emit(recs, len(vidx), n, variant_idx=vidx).reshape(len(vidx), -1)
    assert body == b"".join(p + s.tobytes() for p, s in zip(prefixes, seg))


def test_emit_lines_oracle():
    rng = np.random.default_rng(3)
    n, v = 19, 6
    r = oracle.variant_record_size(n)
    recs = rng.integers(0, 256, size=v * r, dtype=np.uint8)
    prefixes = [b"1\t%d\trs%d\tA\tC\t.\t.\t.\tGT" % (i * 1000, i) for i in range(v)]
    blob = np.frombuffer(b"".join(prefixes), dtype=np.uint8)
    poff = np.cumsum([0] + [len(p) for p in prefixes]).astype(np.uint64)
    loff = np.cumsum([0] + [len(p) + 4 * n + 1 for p in prefixes]).astype(np.uint64)
    got = oracle.emit_lines(recs, v, n, blob, poff, loff)
    seg = oracle.decode_emit(recs, v, n).reshape(v, -1)
    assert got.tobytes() == b"".join(p + s.tobytes() for p, s in zip(prefixes, seg))


def test_hwe_distribution_is_what_it_says():
    """SURVEY.md §8d "hwe": per-variant allele frequency p in [0.01, 0.5), genotype codes in Hardy-Weinberg proportions
    (q^2, 2pq, p^2) and 0.1 % missing.  Checked on the oracle's generator (the device twin is compared byte for byte in the
    GPU leg): every variant's observed allele frequency matches its p16, heterozygosity matches 2pq, missing rate 0.1 %."""
    n, v = 200_000, 6
    recs = oracle.synth_records(n, v, first_variant=1234, hwe=True).reshape(v, -1)
    codes = ((recs[:, :, None] >> np.array([0, 2, 4, 6], dtype=np.uint8)) & 3).reshape(v, -1)[:, :n]
    for j in range(v):
        p = (655 + oracle.splitmix64((0x5047454E ^ 0x4D4146) + 1234 + j) % 32113) / 65536.0
        assert 0.0099 < p < 0.5
        c = codes[j]
        called = c[c != 3]
        assert abs((c == 3).mean() - 0.001) < 0.0004
        af = (called == 1).mean() * 0.5 + (called == 2).mean()
        assert abs(af - p) < 0.004
        assert abs((called == 1).mean() - 2 * p * (1 - p)) < 0.006
    assert (recs[:, -1] >> ((n % 4) * 2 if n % 4 else 8) == 0).all()
