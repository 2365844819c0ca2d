"""Probe: full-line assembly (pgenhip_emit_lines) vs GT-only decode on the chr22 shape."""
import sys, statistics
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
import numpy as np
import torch
import pgen_rs_amd

def main(n=2504, v=1_103_547, rounds=5):
    with pgen_rs_amd.GtEngine(n, device=0) as eng:
        recs = eng.synth_records(v)
        # prefixes shaped like SURVEY config 2's pvar rows: 22\t{pos}\tsnp{i}\tA\tG\t100\tPASS\t.\tGT
        lens = np.array([len(f"22\t{16050000 + 7 * i}\tsnp{i}\tA\tG\t100\tPASS\t.\tGT") for i in range(0, v, 1000)], dtype=np.int64)
        plen = np.repeat(lens, 1000)[:v]
        prefix_off = np.zeros(v + 1, dtype=np.int64); np.cumsum(plen, out=prefix_off[1:])
        line_off = np.zeros(v + 1, dtype=np.int64); np.cumsum(plen + eng.gt_row_bytes, out=line_off[1:])
        blob = torch.full((int(prefix_off[-1]),), 65, dtype=torch.uint8, device="cuda:0")
        d_poff = torch.from_numpy(prefix_off).to("cuda:0"); d_loff = torch.from_numpy(line_off).to("cuda:0")
        out = torch.empty(int(line_off[-1]) + 64, dtype=torch.uint8, device="cuda:0")
        max_prefix = int(plen.max())  # host-side preparation stays outside the timed region
        ts = []
        for r in range(rounds + 1):
            eng.timer_start()
            eng.emit_lines(recs, v, blob, d_poff, d_loff, max_prefix, out)
            ms = eng.timer_stop()
            if r: ts.append(ms)
        tot = int(line_off[-1]) + v * eng.record_size + int(prefix_off[-1])
        med = statistics.median(ts)
        print(f"emit_lines N={n} V={v}: med {med:.3f} ms  {tot/med/1e9:.3f} TB/s (lines {int(line_off[-1])/1e9:.2f} GB)", flush=True)
        out2 = torch.empty(v * eng.gt_row_bytes, dtype=torch.uint8, device="cuda:0")
        ts = []
        for r in range(rounds + 1):
            eng.timer_start(); eng.decode_emit(recs, v, out=out2); ms = eng.timer_stop()
            if r: ts.append(ms)
        med = statistics.median(ts)
        print(f"decode_emit: med {med:.3f} ms  {v*(eng.record_size+eng.gt_row_bytes)/med/1e9:.3f} TB/s", flush=True)

if __name__ == "__main__":
    main()
