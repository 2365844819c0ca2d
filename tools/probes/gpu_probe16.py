"""Probe: stream-kernel knobs, interleaved in ONE process (see gpu_probe15.py for why)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, pgen_rs_amd
KEYS = ("PGENHIP_WIDE_BURST", "PGENHIP_WIDE_NT", "PGENHIP_WIDE_RANGES", "PGENHIP_WIDE_BLOCKS_PER_CU", "PGENHIP_USE_SPAN", "PGENHIP_WIDE_DYN", "PGENHIP_WIDE_STREAM")
SETTINGS = {
    "default": {},
    "burst2": {"PGENHIP_WIDE_BURST": "2"},
    "burst4": {"PGENHIP_WIDE_BURST": "4"},
    "nt0": {"PGENHIP_WIDE_NT": "0"},
    "bpc3": {"PGENHIP_WIDE_BLOCKS_PER_CU": "3"},
    "span": {"PGENHIP_USE_SPAN": "1"},
    "span nt0": {"PGENHIP_USE_SPAN": "1", "PGENHIP_WIDE_NT": "0"},
    "static7": {"PGENHIP_WIDE_DYN": "0"},
}
def main(n=2504, v=1_103_547, rounds=7):
    with pgen_rs_amd.GtEngine(n, device=0) as eng:
        recs = eng.synth_records(v)
        out = torch.empty(v * eng.gt_row_bytes, dtype=torch.uint8, device="cuda:0")
        ts = {k: [] for k in SETTINGS}
        for i in range(rounds + 1):
            for name, env in SETTINGS.items():
                for k in KEYS: os.environ.pop(k, None)
                os.environ.update(env)
                for _ in range(2):
                    eng.timer_start(); eng.decode_emit(recs, v, out=out); ms = eng.timer_stop()
                if i: ts[name].append(ms)
        for name, x in ts.items():
            print(f"{name:10s}: med {statistics.median(x):.3f} min {min(x):.3f} max {max(x):.3f} ms", flush=True)
if __name__ == "__main__":
    main()
