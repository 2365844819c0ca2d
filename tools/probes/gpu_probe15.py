"""Probe: work-queue stream kernel, number of item ranges (write fronts), interleaved IN ONE PROCESS
(process-to-process placement of the buffers moves the time by +-5 %, so separate runs cannot be compared)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, pgen_rs_amd

def main(n=2504, v=1_103_547, rounds=9):
    with pgen_rs_amd.GtEngine(n, device=0) as eng:
        recs = eng.synth_records(v)
        out = torch.empty(v * eng.gt_row_bytes, dtype=torch.uint8, device="cuda:0")
        ts = {r: [] for r in (8, 4, 2, 1)}
        for i in range(rounds + 1):
            for r in ts:
                os.environ["PGENHIP_WIDE_RANGES"] = str(r)
                for _ in range(2):
                    eng.timer_start(); eng.decode_emit(recs, v, out=out); ms = eng.timer_stop()
                if i: ts[r].append(ms)
        for r, x in ts.items():
            print(f"ranges={r}: med {statistics.median(x):.3f} min {min(x):.3f} max {max(x):.3f} ms", flush=True)

if __name__ == "__main__":
    main()
