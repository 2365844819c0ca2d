set -o pipefail
mkdir -p gpurun_out/r3o
timeout -k 10 1000 python -m pytest tests/test_host_cli_gpu.py tests/test_host_cli_fullsize_gpu.py -x -q -m gpu > gpurun_out/r3o/tests.log 2>&1 || { tail -40 gpurun_out/r3o/tests.log; exit 1; }
tail -3 gpurun_out/r3o/tests.log
bash tools/e2e_chr22.sh > gpurun_out/r3o/e2e.log 2>&1; cat gpurun_out/r3o/e2e.log
