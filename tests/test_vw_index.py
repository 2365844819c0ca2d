"""Variable-width .pgen storage modes, header / offset-table walk — CPU leg.

Three readings of the same tables must agree: the numpy WRITER's bookkeeping (tests/golden/make_golden_vw.py, committed
fixtures), the C oracle (oracle/pgen_vw_oracle.c: a literal restatement of the reference's `Pgen` validator,
/root/reference/src/pgen.rs, plus the per-variant index the reference never computes) and the product's host-side walk
behind the C ABI (pgenhip_vw_*).  PARITY UNPINNED: the reference has no file, test or output for these modes."""
import hashlib
import json
import sys
from pathlib import Path

import numpy as np
import pytest

import pgen_oracle as oracle
import pgen_rs_amd
from pgen_rs_amd import _capi

GOLDEN = Path(__file__).resolve().parent / "golden"
sys.path.insert(0, str(GOLDEN))
import make_golden_vw as writer  # noqa: E402  (the committed generator: used here for the cases too big to commit)

INDEX = json.loads((GOLDEN / "vw" / "index.json").read_text())


def both_walks(data: bytes):
    rc, oh = oracle.vw_parse_header(data[:12])
    assert rc == 0
    orc, ot, ol, oo = oracle.vw_index(oh, data)
    ph = pgen_rs_amd.vw_parse_header(data[:12])
    pt, pl, po = pgen_rs_amd.vw_walk_index(ph, data[12 : ph.variant_records_offset])
    return (oh, orc, ot, ol, oo), (ph, pt, pl, po)


@pytest.mark.parametrize("name", sorted(INDEX))
def test_fixture_tables_oracle_and_product_agree(name):
    exp = INDEX[name]
    data = (GOLDEN / "vw" / f"{name}.pgen").read_bytes()
    assert hashlib.sha256(data).hexdigest() == exp["file_sha256"]
    (oh, orc, ot, ol, oo), (ph, pt, pl, po) = both_walks(data)
    assert orc == 0
    for h in (oh, ph):
        assert (h.variant_count, h.sample_count, h.record_type_bits, h.record_length_bytes, h.storage_mode) == \
            (exp["variant_count"], exp["sample_count"], exp["record_type_bits"], exp["record_length_bytes"], exp["storage_mode"])
    geo = oracle.vw_geometry(oh)
    assert geo["block_count"] == ph.block_count == exp["block_count"]
    assert geo["main_header_body_offset"] == ph.main_header_body_offset == 12 + 8 * exp["block_count"]
    assert geo["variant_records_offset"] == ph.variant_records_offset == exp["variant_records_offset"]
    for types, lens, offs in ((ot, ol, oo), (pt, pl, po)):
        assert types.tolist() == exp["types"] and lens.tolist() == exp["lens"] and offs.tolist() == exp["offs"]
    # the reference's own validation walk (src/pgen.rs:140-258) lands where the header says the records start
    p1, p2, types_seen, _len_bytes = oracle.vw_validate(oh, data)
    assert p1 == geo["main_header_body_offset"] and p2 == geo["variant_records_offset"]
    assert set(types_seen) >= set(exp["types"])  # (4-bit: the padding nibble of an odd count adds a 0)


@pytest.mark.parametrize("name", sorted(INDEX))
def test_type0_records_decode_to_the_fixture_text(name):
    exp = INDEX[name]
    data = (GOLDEN / "vw" / f"{name}.pgen").read_bytes()
    n = exp["sample_count"]
    r = oracle.variant_record_size(n)
    sel = [i for i, t in enumerate(exp["types"]) if t == 0]
    ph = pgen_rs_amd.vw_parse_header(data[:12])
    pt, pl, po = pgen_rs_amd.vw_walk_index(ph, data[12 : ph.variant_records_offset])
    offs = pgen_rs_amd.vw_select_uncompressed(pt, pl, po, r, sel)
    assert offs.tolist() == [exp["offs"][i] for i in sel]
    got = oracle.decode_emit_at(np.frombuffer(data, dtype=np.uint8), offs, n)
    want = (GOLDEN / "vw" / f"{name}.gt").read_bytes()
    assert hashlib.sha256(want).hexdigest() == exp["gt_type0_sha256"]
    assert got.tobytes() == want


def test_compressed_record_is_reported_not_guessed():
    exp = INDEX["mixed_8bit_len2"]
    data = (GOLDEN / "vw" / "mixed_8bit_len2.pgen").read_bytes()
    ph = pgen_rs_amd.vw_parse_header(data[:12])
    pt, pl, po = pgen_rs_amd.vw_walk_index(ph, data[12 : ph.variant_records_offset])
    first_bad = next(i for i, t in enumerate(exp["types"]) if t != 0)
    with pytest.raises(pgen_rs_amd.PgenHipError) as ei:
        pgen_rs_amd.vw_select_uncompressed(pt, pl, po, 76)  # all variants
    assert ei.value.status == _capi.ERR_COMPRESSED_RECORD and f"variant {first_bad}:" in str(ei.value)
    with pytest.raises(pgen_rs_amd.PgenHipError) as ei:   # type 0 but a wrong length is not a plain record either
        pgen_rs_amd.vw_select_uncompressed(pt, pl, po, 75, [0])
    assert ei.value.status == _capi.ERR_COMPRESSED_RECORD
    with pytest.raises(pgen_rs_amd.PgenHipError) as ei:
        pgen_rs_amd.vw_select_uncompressed(pt, pl, po, 76, [len(exp["types"])])
    assert ei.value.status == _capi.ERR_INDEX_RANGE


def test_header_asserts_of_the_reference():
    good = bytes([0x6C, 0x1B, 0x10]) + (9).to_bytes(4, "little") + (11).to_bytes(4, "little")
    assert oracle.vw_parse_header(good + bytes([0x40]))[0] == 0
    for fmt, orc in ((0x00, -2), (0x80, -2), (0xC0, -2), (0x48, -3), (0x4F, -3)):   # provisional ref (:58), record storage mode (:64)
        assert oracle.vw_parse_header(good + bytes([fmt]))[0] == orc
        with pytest.raises(pgen_rs_amd.PgenHipError) as ei:
            pgen_rs_amd.vw_parse_header(good + bytes([fmt]))
        assert ei.value.status == _capi.ERR_BAD_FLAGS
    assert oracle.vw_parse_header(b"\x6c\x1c" + good[2:] + b"\x40")[0] == -1     # :30
    with pytest.raises(pgen_rs_amd.PgenHipError) as ei:
        pgen_rs_amd.vw_parse_header(b"\x6c\x1c" + good[2:] + b"\x40")
    assert ei.value.status == _capi.ERR_BAD_MAGIC
    with pytest.raises(pgen_rs_amd.PgenHipError) as ei:      # allele-count arrays: outside the slice (the reference sizes the tables without them, :116-133)
        pgen_rs_amd.vw_parse_header(good + bytes([0x50]))
    assert ei.value.status == _capi.ERR_BAD_FLAGS
    for mode in range(8):   # :61-67
        rc, h = oracle.vw_parse_header(good + bytes([0x40 | mode]))
        ph = pgen_rs_amd.vw_parse_header(good + bytes([0x40 | mode]))
        assert rc == 0 and h.record_type_bits == ph.record_type_bits == (4 if mode < 4 else 8)
        assert h.record_length_bytes == ph.record_length_bytes == mode % 4 + 1


@pytest.mark.parametrize("v,tb,lb", [(70_000, 4, 1), (65_537, 8, 2), (131_073, 4, 2)])
def test_several_blocks(v, tb, lb):
    """More than one 65 536-variant block (too big to commit: generated here by the committed writer): block offsets, the
    per-block rounding of 4-bit type arrays, a gap in front of every block's records."""
    rng = np.random.default_rng(v)
    n = 7  # R = 2
    types = np.where(rng.random(v) < 0.9, 0, rng.integers(1, 8, size=v)).tolist()
    recs = writer.make_records(rng, n, types)
    data, exp = writer.write_vw(n, recs, tb, lb, block_gap=9)
    (oh, orc, ot, ol, oo), (ph, pt, pl, po) = both_walks(data)
    assert orc == 0 and ph.block_count == (v + 65535) // 65536
    for types_, lens, offs in ((ot, ol, oo), (pt, pl, po)):
        assert types_.tolist() == exp["types"] and lens.tolist() == exp["lens"] and offs.tolist() == exp["offs"]
    p1, p2, _, _ = oracle.vw_validate(oh, data)
    assert p1 == ph.main_header_body_offset and p2 == ph.variant_records_offset == exp["variant_records_offset"]
    sel = [i for i, t in enumerate(exp["types"]) if t == 0][:: max(1, v // 500)]
    offs = pgen_rs_amd.vw_select_uncompressed(pt, pl, po, 2, sel)
    got = oracle.decode_emit_at(np.frombuffer(data, dtype=np.uint8), offs, n)
    want = writer.gt_of_type0(n, [recs[i] for i in sel])
    assert got.tobytes() == want


def test_reference_last_block_quirk_is_documented():
    """src/pgen.rs:200-204 takes variant_count % 65 536 for the last block — zero when the count is a multiple of the block
    size, so the reference's own walk stops one block short and its assert_eq at :95 would fire.  The oracle keeps the quirk
    (literal restatement); the product reads the last block as full."""
    rng = np.random.default_rng(5)
    v, n = 65_536, 3
    recs = writer.make_records(rng, n, [0] * v)
    data, exp = writer.write_vw(n, recs, 8, 1)
    (oh, orc, ot, ol, oo), (ph, pt, pl, po) = both_walks(data)
    p1, p2, _, _ = oracle.vw_validate(oh, data)
    assert p1 == ph.main_header_body_offset
    assert p2 == p1 and p2 != exp["variant_records_offset"]          # the reference's walk: zero variants in the last block
    assert ph.variant_records_offset == exp["variant_records_offset"]  # the product: 65 536
    assert orc == 0 and oo.tolist() == po.tolist() == exp["offs"]


def test_bad_tables_are_refused():
    rng = np.random.default_rng(6)
    v, n = 70_000, 7
    recs = writer.make_records(rng, n, [0] * v)
    data, _ = writer.write_vw(n, recs, 8, 1)
    ph = pgen_rs_amd.vw_parse_header(data[:12])
    # block offsets swapped: not ascending (src/pgen.rs:160-165 panics)
    bad = bytearray(data)
    bad[12:20], bad[20:28] = data[20:28], data[12:20]
    rc, oh = oracle.vw_parse_header(bytes(bad[:12]))
    assert oracle.vw_validate(oh, bytes(bad))[0] == -2 and oracle.vw_index(oh, bytes(bad))[0] == -2
    with pytest.raises(pgen_rs_amd.PgenHipError) as ei:
        pgen_rs_amd.vw_walk_index(ph, bytes(bad[12 : ph.variant_records_offset]))
    assert ei.value.status == _capi.ERR_BAD_INDEX
    # truncated tables
    with pytest.raises(pgen_rs_amd.PgenHipError) as ei:
        pgen_rs_amd.vw_walk_index(ph, data[12 : ph.variant_records_offset - 1])
    assert ei.value.status == _capi.ERR_BAD_INDEX
    # second block's offset inside the first block's records
    bad = bytearray(data)
    first = int.from_bytes(data[12:20], "little")
    bad[20:28] = (first + 5).to_bytes(8, "little")
    with pytest.raises(pgen_rs_amd.PgenHipError) as ei:
        pgen_rs_amd.vw_walk_index(ph, bytes(bad[12 : ph.variant_records_offset]))
    assert ei.value.status == _capi.ERR_BAD_INDEX
    assert oracle.vw_index(oh, bytes(bad))[0] == -3
