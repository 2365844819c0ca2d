"""Diagnostic (test infrastructure): one decode at a big shape with the wide-kernel knobs taken from the
environment, rows spot-checked against the oracle.  usage: cfg3_check.py V N"""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "oracle")
import torch
import pgen_rs_amd, pgen_oracle as oracle
v, n = int(sys.argv[1]), int(sys.argv[2])
r = (2 * n + 7) // 8
row = 4 * n + 1
with pgen_rs_amd.GtEngine(n, device=0) as eng:
    recs = eng.synth_records(v); eng.wait()
    print("synth ok", flush=True)
    out = eng.decode_emit(recs, v, kernel=4); eng.wait()
    print("decode ok", flush=True)
    assert bool((out[row - 1 :: row] == 10).all())
    for j in (0, 1, v // 2, v - 1):
        got = out[j * row : (j + 1) * row].cpu().numpy()
        host = recs[j * r : (j + 1) * r].cpu().numpy()
        assert got.tobytes() == oracle.decode_emit(host, 1, n).tobytes(), j
print("rows ok")
