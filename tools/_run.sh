set -o pipefail
mkdir -p gpurun_out/r3n
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r3n/tests.log 2>&1 || { tail -40 gpurun_out/r3n/tests.log; exit 1; }
tail -3 gpurun_out/r3n/tests.log
