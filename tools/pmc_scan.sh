#!/bin/bash
# PMC passes for the kept-subset path at config-5 geometry (N = 500 000, 1 % kept).
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-pmc_scan}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--samples 500000 --variants 16000 --keep-modulus ${2:-100} --steps 3 --warmup 1 --no-cpu-baseline --no-host-delivered --no-secondary --kernel ${3:-3}"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d $OUT/p1 -- python3 $ROOT/bench.py $ARGS > $OUT/p1.log 2>&1 || { tail -5 $OUT/p1.log; exit 1; }
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM --kernel-trace --output-format csv -d $OUT/p2 -- python3 $ROOT/bench.py $ARGS > $OUT/p2.log 2>&1 || { tail -5 $OUT/p2.log; exit 1; }
python3 - <<PY
import csv, glob, collections
for d in ("p1", "p2"):
    f = glob.glob("$OUT/%s/*/*_counter_collection.csv" % d)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "gt_scan" in r["Kernel_Name"] or "gt_rows" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k, v in sorted(agg.items()):
        print(d, k[0], k[1], "mean %.4g" % (sum(v) / len(v)), "n", len(v))
PY
