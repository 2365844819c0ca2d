"""Probe: PCIe-inclusive ("host-delivered") rate of the hot path on the chr22 shape.

Records start in pinned host memory and the GT text ends in pinned host memory; blocks of
variants go H2D -> pgenhip_decode_emit -> D2H on two streams (double buffering), as the CLI does
without its file I/O.  SURVEY 8(d): report kernel-resident AND host-delivered rates; `bench.py`'s
`value` is the kernel-resident one.
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import pgen_rs_amd


def main(n=2504, v=1_103_547, block=65_536):
    dev = "cuda:0"
    with pgen_rs_amd.GtEngine(n, device=0) as e0, pgen_rs_amd.GtEngine(n, device=0) as e1:
        engs = (e0, e1)
        r, row = e0.record_size, e0.gt_row_bytes
        h_recs = torch.empty(v * r, dtype=torch.uint8).pin_memory()
        h_recs.copy_(e0.synth_records(v).cpu())
        h_out = torch.empty(v * row, dtype=torch.uint8).pin_memory()
        streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
        d_recs = [torch.empty(block * r, dtype=torch.uint8, device=dev) for _ in range(2)]
        d_out = [torch.empty(block * row, dtype=torch.uint8, device=dev) for _ in range(2)]
        for s, e in zip(streams, engs):
            with torch.cuda.stream(s):
                e.use_torch_stream()  # binds the ctx to torch's current stream = s

        def run():
            for i, b0 in enumerate(range(0, v, block)):
                nb = min(block, v - b0)
                k = i & 1
                with torch.cuda.stream(streams[k]):
                    d_recs[k][: nb * r].copy_(h_recs[b0 * r : (b0 + nb) * r], non_blocking=True)
                    engs[k].decode_emit(d_recs[k], nb, out=d_out[k])
                    h_out[b0 * row : (b0 + nb) * row].copy_(d_out[k][: nb * row], non_blocking=True)
            torch.cuda.synchronize()

        run()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); run(); ts.append(time.perf_counter() - t0)
        t = min(ts)
        assert int(h_out[-1]) == 10 and int(h_out[0]) == 9
        print(f"host-delivered chr22 shape: {t*1e3:.1f} ms  {v*n/t:.3e} genotypes/s  D2H {v*row/t/1e9:.1f} GB/s  H2D {v*r/t/1e9:.1f} GB/s", flush=True)


if __name__ == "__main__":
    main()
