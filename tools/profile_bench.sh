#!/bin/bash
# Runs on the GPU box (via gpurun): plain bench line, rocprofv3 kernel-trace stats, and the two
# PMC passes (FETCH_SIZE / WRITE_SIZE need separate runs: TCC has 4 counter slots); then the
# summary files that get committed under profiles/<tag>/ (tools/summarize_profile.py).
# usage: tools/profile_bench.sh <tag> [bench args...]
set -o pipefail
set -u
TAG=${1:?usage: profile_bench.sh <tag> [bench args...]}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT="$ROOT/gpurun_out/$TAG"
rm -rf -- "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/bench.py "$@" > $OUT/bench.json 2> $OUT/bench.err || { echo "bench failed"; tail -5 $OUT/bench.err; exit 1; }
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-host-delivered --no-secondary "$@" > $OUT/trace.log 2>&1 || { echo "trace failed"; tail -5 $OUT/trace.log; exit 1; }
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-host-delivered --no-secondary "$@" > $OUT/pmc_fetch.log 2>&1 || { echo "pmc fetch failed"; tail -5 $OUT/pmc_fetch.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-host-delivered --no-secondary "$@" > $OUT/pmc_write.log 2>&1 || { echo "pmc write failed"; tail -5 $OUT/pmc_write.log; exit 1; }
python3 $ROOT/tools/summarize_profile.py $OUT $OUT/summary
# keep the merge-back small: the raw traces are large, the summary is what gets committed
rm -rf $OUT/trace/*/*_kernel_trace.csv $OUT/pmc_fetch/*/*_kernel_trace.csv $OUT/pmc_write/*/*_kernel_trace.csv
