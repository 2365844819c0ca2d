#!/bin/bash
# does the measured rate depend on how long the GPU has been busy?  (clock ramp / power state)
run() { python bench.py --no-cpu-baseline --steps $2 --warmup $1 > gpurun_out/sw.json 2>/dev/null && python -c "import json;d=json.load(open('gpurun_out/sw.json'));print('warmup $1 steps $2:', round(d['ms_per_step'],3), round(d['roofline']['frac'],3))"; }
for rep in 1 2 3; do
run 5 30
run 300 100
run 1000 200
sleep 20
done
