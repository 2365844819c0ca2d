"""Probe: is the allocation-placement effect (gpu_probe17.py) a property of the memory, i.e. does a plain fill see it too?"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, pgen_rs_amd

def timed(fn, rounds=6):
    ts = []
    for i in range(rounds + 2):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); e.synchronize()
        if i >= 2: ts.append(s.elapsed_time(e))
    return statistics.median(ts)

def main(n=2504, v=1_103_547):
    with pgen_rs_amd.GtEngine(n, device=0) as eng:
        recs = eng.synth_records(v)
        keep = []
        pattern = torch.arange(0, 1 << 20, dtype=torch.int32, device="cuda:0")
        for k in range(8):
            out = torch.empty(v * eng.gt_row_bytes, dtype=torch.uint8, device="cuda:0")
            keep.append(out)
            t_gt = timed(lambda: eng.decode_emit(recs, v, out=out))
            t_fill = timed(lambda: out.fill_(7))
            o32 = out[: (out.numel() // 4) * 4].view(torch.int32)
            t_read = timed(lambda: o32.sum())
            print(f"allocation {k} at {out.data_ptr():#x}: GT kernel {t_gt:.3f} ms   torch fill {t_fill:.3f} ms ({out.numel()/t_fill/1e9:.2f} TB/s)   torch sum(read) {t_read:.3f} ms", flush=True)

if __name__ == "__main__":
    main()
