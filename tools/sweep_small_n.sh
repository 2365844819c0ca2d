#!/bin/bash
# all samples kept on short rows: which kernel?  (bench.py --kernel: 2 flat, 4 stream (wide), 6 pick with the identity)
run() { python bench.py --samples $1 --variants $((2760000000 / $1)) --steps 10 --warmup 2 --no-cpu-baseline --kernel $2 > gpurun_out/sw.json 2>/dev/null && python -c "import json;d=json.load(open('gpurun_out/sw.json'));print('N=$1 kernel $2:', round(d['ms_per_step'],3), round(d['roofline']['frac'],3))" || echo "N=$1 kernel $2: n/a"; }
for n in 100 500 1000; do run $n 2; run $n 6; done
for n in 1024 1500 2000 2504 4096; do run $n 4; run $n 6; done
