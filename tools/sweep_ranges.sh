#!/bin/bash
# work-queue stream kernel: how many contiguous item ranges (write fronts)?  chr22 block
run() { python bench.py --no-cpu-baseline --steps 40 > gpurun_out/sw.json 2>/dev/null && python -c "import json;d=json.load(open('gpurun_out/sw.json'));print('$1', round(d['ms_per_step'],3), round(d['roofline']['frac'],3))"; }
for rep in 1 2 3; do
for r in 8 4 2 1; do PGENHIP_WIDE_RANGES=$r run "ranges=$r"; done
done
