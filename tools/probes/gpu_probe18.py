"""Debug: HIP graph capture + replay of one pgenhip_decode_emit (work-queue stream kernel)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, pgen_rs_amd
DEV = "cuda:0"
n, v = 2504, 257
with pgen_rs_amd.GtEngine(n, device=0) as eng:
    r = eng.record_size
    d_recs = torch.zeros(v * r, dtype=torch.uint8, device=DEV)
    out = torch.full((v * eng.gt_row_bytes,), 0xA5, dtype=torch.uint8, device=DEV)
    side = torch.cuda.Stream(device=DEV)
    with torch.cuda.stream(side):
        eng.use_torch_stream()
        eng.decode_emit(d_recs, v, out=out)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        eng.use_torch_stream()
        eng.decode_emit(d_recs, v, out=out)
    print("captured", flush=True)
    for rep in range(4):
        out.fill_(0xA5)
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        o = out.cpu().numpy()
        print("replay", rep, "untouched bytes:", int((o == 0xA5).sum()), "of", o.size, "first bytes", bytes(o[:8]), flush=True)
    try:
        g.debug_dump("/tmp/graph_dump")
    except Exception as e:
        print("no dump:", e)
