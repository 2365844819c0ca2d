#!/bin/bash
# Counter groups (one rocprofv3 --pmc pass each) around tools/ab_probe.py; prints the mean per kernel and counter.
# usage (GPU box): tools/pmc_groups.sh <tag> "<group1 counters>;<group2 counters>;..." <ab_probe args...>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:?tag}; GROUPS_=${2:?groups}; shift 2
OUT="$ROOT/gpurun_out/$TAG"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
IFS=';' read -ra GS <<< "$GROUPS_"
n=0
for G in "${GS[@]}"; do
  n=$((n+1))
  rocprofv3 --pmc $G --kernel-trace --output-format csv -d $OUT/g$n -- python3 $ROOT/tools/ab_probe.py --rounds 1 --reps 1 "$@" > $OUT/g$n.log 2>&1 || echo "pass '$G' failed: $(tail -1 $OUT/g$n.log)"
  rm -f $OUT/g$n/*/*kernel_trace.csv
done
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/g*/*/*_counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "pgenhip" in r["Kernel_Name"] and "synth" not in r["Kernel_Name"]:
            agg[(r["Kernel_Name"].split("<")[0].split("::")[-1][-28:], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k, v in sorted(agg.items()):
        print(k[0], k[1], "mean %.6g" % (sum(v) / len(v)), "n", len(v))
PY
