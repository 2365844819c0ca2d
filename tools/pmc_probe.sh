#!/bin/bash
# SQ counter passes around tools/ab_probe.py for any shape / arm: where a kernel's wave cycles go (busy, waiting, VALU, LDS) and how many
# instructions of each kind it issues.  usage (GPU box): tools/pmc_probe.sh <tag> <ab_probe args...>
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:?tag}; shift
OUT="$ROOT/gpurun_out/$TAG"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--rounds 2 --reps 2 $*"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $OUT/p1 -- python3 $ROOT/tools/ab_probe.py $ARGS > $OUT/p1.log 2>&1 || { tail -5 $OUT/p1.log; exit 1; }
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_INST_LEVEL_VMEM --kernel-trace --output-format csv -d $OUT/p2 -- python3 $ROOT/tools/ab_probe.py $ARGS > $OUT/p2.log 2>&1 || { tail -5 $OUT/p2.log; exit 1; }
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_FLAT --kernel-trace --output-format csv -d $OUT/p3 -- python3 $ROOT/tools/ab_probe.py $ARGS > $OUT/p3.log 2>&1 || { tail -5 $OUT/p3.log; echo "(p3 failed: counters not all available)"; }
tail -2 $OUT/p1.log
python3 - <<PY
import csv, glob, collections
for d in ("p1", "p2", "p3"):
    fs = glob.glob("$OUT/%s/*/*_counter_collection.csv" % d)
    if not fs: continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "pgenhip" in r["Kernel_Name"] and "synth" not in r["Kernel_Name"]:
            agg[(r["Kernel_Name"].split("(")[0][-34:], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k, v in sorted(agg.items()):
        print(d, k[0], k[1], "mean %.5g" % (sum(v) / len(v)), "n", len(v))
PY
rm -rf $OUT/p1/*/*kernel_trace.csv $OUT/p2/*/*kernel_trace.csv $OUT/p3/*/*kernel_trace.csv
