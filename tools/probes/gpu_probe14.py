"""Probe: does the stream kernel's time depend on where the output / record buffers sit (channel phase of the write fronts)?"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, pgen_rs_amd

def main(n=2504, v=1_103_547):
    with pgen_rs_amd.GtEngine(n, device=0) as eng:
        recs_big = torch.empty(v * eng.record_size + (64 << 20), dtype=torch.uint8, device="cuda:0")
        recs0 = eng.synth_records(v)
        out_big = torch.empty(v * eng.gt_row_bytes + (512 << 20), dtype=torch.uint8, device="cuda:0")
        print("base addresses: out %#x recs %#x" % (out_big.data_ptr(), recs_big.data_ptr()), flush=True)
        def t(out_off, rec_off, rounds=8):
            r = recs_big[rec_off : rec_off + recs0.numel()]
            r.copy_(recs0)
            o = out_big[out_off : out_off + v * eng.gt_row_bytes]
            ts = []
            for i in range(rounds + 2):
                eng.timer_start(); eng.decode_emit(r, v, out=o); ms = eng.timer_stop()
                if i >= 2: ts.append(ms)
            return statistics.median(ts)
        for out_off in (0, 128, 4096, 65536, 1 << 20, 3 << 20, 16 << 20, 33 << 20, 128 << 20, 257 << 20):
            print(f"out +{out_off:>10}: {t(out_off, 0):.3f} ms", flush=True)
        for rec_off in (0, 4096, 1 << 20, 7 << 20, 32 << 20):
            print(f"recs +{rec_off:>10}: {t(0, rec_off):.3f} ms", flush=True)

if __name__ == "__main__":
    main()
