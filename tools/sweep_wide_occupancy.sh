#!/bin/bash
run() { python bench.py --no-cpu-baseline --steps 40 > gpurun_out/sw.json 2>/dev/null && python -c "import json;d=json.load(open('gpurun_out/sw.json'));print('$1', round(d['ms_per_step'],3), round(d['roofline']['frac'],3))"; }
for rep in 1 2; do
for b in 1 2 3 4; do PGENHIP_WIDE_BLOCKS_PER_CU=$b run "dyn7 bpc=$b"; done
for b in 1 2 3 4 6 8; do PGENHIP_WIDE_STREAM=3 PGENHIP_WIDE_DYN=0 PGENHIP_WIDE_BLOCKS_PER_CU=$b run "static3 bpc=$b"; done
for b in 1 2 3; do PGENHIP_WIDE_STREAM=7 PGENHIP_WIDE_DYN=0 PGENHIP_WIDE_BLOCKS_PER_CU=$b run "static7 bpc=$b"; done
done
