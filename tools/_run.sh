set -o pipefail
mkdir -p gpurun_out/r3s
timeout -k 10 900 python -m pytest tests/test_gt_parity_gpu.py tests/test_host_cli_gpu.py -x -q -m gpu -k "lines or randomized or basic2 or tiny_keep" > gpurun_out/r3s/tests.log 2>&1 || { tail -30 gpurun_out/r3s/tests.log; exit 1; }
tail -3 gpurun_out/r3s/tests.log
for spec in "100 20000000 30" "300 8000000 30" "1000 2500000 30" "300 8000000 166" "100 20000000 10"; do set -- $spec; python tools/ab_probe.py --samples $1 --variants $2 --lines $3 --arms auto --rounds 5 2>&1 | tail -2; done > gpurun_out/r3s/ab.log; cat gpurun_out/r3s/ab.log
python tools/ab_probe.py --samples 300 --variants 8000000 --lines 30 --keep-frac 0.5 --arms auto --rounds 5 2>&1 | tail -2 >> gpurun_out/r3s/ab.log; tail -2 gpurun_out/r3s/ab.log
