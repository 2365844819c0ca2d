#!/bin/bash
# row-item work-queue kernel vs stream-span kernel (both one resident round from the occupancy API), chr22 block
run() { python bench.py --no-cpu-baseline --steps 40 > gpurun_out/sw.json 2>/dev/null && python -c "import json;d=json.load(open('gpurun_out/sw.json'));print('$1', round(d['ms_per_step'],3), round(d['roofline']['frac'],3))"; }
for rep in 1 2 3; do
run "rows"
PGENHIP_USE_SPAN=1 run "span"
PGENHIP_USE_SPAN=1 PGENHIP_WIDE_BLOCKS_PER_CU=3 run "span bpc=3"
done
