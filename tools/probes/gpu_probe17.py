"""Probe: does the stream kernel's time depend on WHICH allocation the output lives in (physical placement), inside one process?"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, pgen_rs_amd

def main(n=2504, v=1_103_547):
    with pgen_rs_amd.GtEngine(n, device=0) as eng:
        recs = eng.synth_records(v)
        keep = []
        for k in range(8):
            out = torch.empty(v * eng.gt_row_bytes, dtype=torch.uint8, device="cuda:0")
            keep.append(out)  # keep every buffer alive: each one is a fresh piece of HBM
            ts = []
            for i in range(8):
                eng.timer_start(); eng.decode_emit(recs, v, out=out); ms = eng.timer_stop()
                if i >= 2: ts.append(ms)
            print(f"output allocation {k} at {out.data_ptr():#x}: {statistics.median(ts):.3f} ms", flush=True)
        # and a fresh copy of the records with the first output buffer
        for k in range(4):
            r2 = recs.clone(); keep.append(r2)
            ts = []
            for i in range(8):
                eng.timer_start(); eng.decode_emit(r2, v, out=keep[0]); ms = eng.timer_stop()
                if i >= 2: ts.append(ms)
            print(f"record copy {k} at {r2.data_ptr():#x}: {statistics.median(ts):.3f} ms", flush=True)

if __name__ == "__main__":
    main()
