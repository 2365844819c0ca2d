#!/bin/bash
# rocprofv3 kernel stats of whole CLI runs (the program directly behind `--`): which kernels a `pgen-hip filter` launches and for how long.
# usage (on the GPU box): tools/profile_cli.sh <variants> <samples> [filter args...]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
V=$1; N=$2; shift 2
D=/dev/shm/pgenhip_prof_$$
mkdir -p $D; trap "rm -rf $D" EXIT
CLI=$ROOT/pgen_rs_amd/pgen-hip
$CLI synth $D/p --variants $V --samples $N > /dev/null || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $D/prof -- $CLI filter $D/p -o $D/out.vcf --stats "$@" 2>/dev/null | tail -1
python3 - $D/prof <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        name = r["Name"].replace("void pgenhip::(anonymous namespace)::", "").replace("pgenhip::(anonymous namespace)::", "")[:60]
        print(f"   {name:60s} calls {r['Calls']:>5s}  total {float(r['TotalDurationNs'])/1e6:9.3f} ms  avg {float(r['AverageNs'])/1e3:9.1f} us")
PY
