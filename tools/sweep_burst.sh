#!/bin/bash
run() { python bench.py --no-cpu-baseline --steps 40 > gpurun_out/sw.json 2>/dev/null && python -c "import json;d=json.load(open('gpurun_out/sw.json'));print('$1', round(d['ms_per_step'],3), round(d['roofline']['frac'],3))"; }
for rep in 1 2 3; do
for b in 1 2 4 8; do PGENHIP_WIDE_BURST=$b run "burst=$b"; done
done
