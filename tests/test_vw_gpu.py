"""Variable-width .pgen storage modes — GPU leg: the file is staged to HBM AS IT LIES ON DISK, the
host-side walk (pgenhip_vw_*) yields the byte offsets of the uncompressed (type-0) records, and every kernel family
decodes them in place through pgenhip_decode_emit_at.  Byte equality with the committed fixtures (numpy writer +
numpy decoder) and with the oracle.  PARITY UNPINNED by the reference (it refuses these modes, src/pfile.rs:53)."""
import json
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

import pgen_oracle as oracle
import pgen_rs_amd
from pgen_rs_amd import _capi

pytestmark = pytest.mark.gpu

GOLDEN = Path(__file__).resolve().parent / "golden"
sys.path.insert(0, str(GOLDEN))
import make_golden_vw as writer  # noqa: E402

INDEX = json.loads((GOLDEN / "vw" / "index.json").read_text())
DEV = "cuda:0"
SENTINEL = 0xA5


def kernels_for(n, subset, k):
    ks = [_capi.KERNEL_AUTO, _capi.KERNEL_ROWS]
    if not subset and n >= 8:
        ks.append(_capi.KERNEL_FLAT)
    if not subset and n >= 1024:
        ks.append(_capi.KERNEL_WIDE)
    if subset and n >= 61:
        ks.append(_capi.KERNEL_SCAN)
    if 61 <= n <= 4096 and (k >= 4 if subset else True):
        ks.append(_capi.KERNEL_PICK)
    return ks


def decode_file(data: bytes, n: int, sel, kept=None):
    """-> {kernel: GT bytes} of the selected variants of the variable-width file `data`, decoded in place on the GPU."""
    r = oracle.variant_record_size(n)
    h = pgen_rs_amd.vw_parse_header(data[:12])
    types, lens, offs = pgen_rs_amd.vw_walk_index(h, data[12 : h.variant_records_offset])
    sel_off = pgen_rs_amd.vw_select_uncompressed(types, lens, offs, r, sel)
    d_file = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).to(DEV)
    d_off = torch.from_numpy(sel_off.astype(np.int64)).to(DEV)
    out = {}
    with pgen_rs_amd.GtEngine(n, kept_idx=kept, device=0) as eng:
        row = eng.gt_row_bytes
        for kern in kernels_for(n, kept is not None, eng.kept_count):
            buf = torch.full((len(sel) * row + 48,), SENTINEL, dtype=torch.uint8, device=DEV)
            eng.decode_emit_at(d_file, d_off, len(sel), out=buf[5:], kernel=kern)
            eng.wait()
            got = buf.cpu().numpy()
            assert (got[:5] == SENTINEL).all() and (got[5 + len(sel) * row :] == SENTINEL).all(), f"kernel {kern} wrote outside its rows"
            out[kern] = got[5 : 5 + len(sel) * row].tobytes()
    return out, sel_off


@pytest.mark.parametrize("name", sorted(INDEX))
def test_fixture_type0_records_in_place(name):
    exp = INDEX[name]
    data = (GOLDEN / "vw" / f"{name}.pgen").read_bytes()
    n = exp["sample_count"]
    sel = [i for i, t in enumerate(exp["types"]) if t == 0]
    want = (GOLDEN / "vw" / f"{name}.gt").read_bytes()
    got, _ = decode_file(data, n, sel)
    for kern, text in got.items():
        assert text == want, f"kernel {kern}"


@pytest.mark.parametrize("n,v,keep_frac", [(2504, 3001, None), (2504, 2003, 0.3), (300, 5000, None), (40_000, 301, 0.02), (70_001, 60, None)])
def test_mixed_file_many_variants(n, v, keep_frac):
    """A file with ~85 % plain records between compressed ones of random length (records at every byte alignment), a
    gapped selection of the plain ones, all samples or a kept subset; against the oracle reading the same bytes."""
    rng = np.random.default_rng(n + v)
    types = np.where(rng.random(v) < 0.85, 0, rng.integers(1, 8, size=v)).tolist()
    recs = writer.make_records(rng, n, types)
    data, exp = writer.write_vw(n, recs, 8, 3, block_gap=1)
    plain = [i for i, t in enumerate(types) if t == 0]
    sel = sorted(rng.choice(plain, size=len(plain) * 3 // 4, replace=False).tolist())
    kept = None if keep_frac is None else np.sort(rng.choice(n, size=int(n * keep_frac), replace=False)).astype(np.uint32)
    got, sel_off = decode_file(data, n, sel, kept)
    want = oracle.decode_emit_at(np.frombuffer(data, dtype=np.uint8), sel_off, n, kept_idx=kept).tobytes()
    for kern, text in got.items():
        assert text == want, f"kernel {kern}"


def test_selecting_a_compressed_record_fails_loudly():
    data = (GOLDEN / "vw" / "mixed_8bit_len2.pgen").read_bytes()
    with pytest.raises(pgen_rs_amd.PgenHipError) as ei:
        decode_file(data, 301, [0, 1, 2])   # variant 2 has record type 1
    assert ei.value.status == _capi.ERR_COMPRESSED_RECORD
