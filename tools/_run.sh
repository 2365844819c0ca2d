set -o pipefail
mkdir -p gpurun_out/r3q
timeout -k 10 900 python -m pytest tests/test_gt_parity_gpu.py tests/test_host_cli_gpu.py -x -q -m gpu -k "lines or cli or host or filter or bgzf or basic or config1 or shards or variable" > gpurun_out/r3q/tests.log 2>&1 || { tail -30 gpurun_out/r3q/tests.log; exit 1; }
tail -3 gpurun_out/r3q/tests.log
( echo "== chr22 shape, everybody"; bash tools/profile_cli.sh 1103547 2504; echo "== chr22 shape, --include-sam KEEP"; bash tools/profile_cli.sh 1103547 2504 --include-sam 'KEEP == "1"'; echo "== 200000 x 300, everybody"; bash tools/profile_cli.sh 200000 300 ) > gpurun_out/r3q/cli_kernels.log 2>&1; grep -v "at::\|rocclr" gpurun_out/r3q/cli_kernels.log
python tools/ab_probe.py --samples 2504 --variants 1103547 --lines 30 --arms auto --rounds 5 > gpurun_out/r3q/ab_lines.log 2>&1; tail -2 gpurun_out/r3q/ab_lines.log
python tools/ab_probe.py --samples 2504 --variants 1103547 --lines 30 --keep-frac 0.1 --arms auto --rounds 5 >> gpurun_out/r3q/ab_lines.log 2>&1; tail -2 gpurun_out/r3q/ab_lines.log
