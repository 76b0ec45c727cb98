#!/bin/bash
# usage: tools/pmc_layer.sh <tag> H Cin Cout op   -> prints per-kernel averages of a few SQ counters
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT -o p1 -- python3 $REPO/tools/one_layer.py "$@" > $OUT/log1.txt 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $OUT -o p2 -- python3 $REPO/tools/one_layer.py "$@" > $OUT/log2.txt 2>&1
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:40]
        if "igemm" not in k: continue
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        print(k, {c: round(sum(v)/len(v)) for c, v in d.items()})
PY
tail -3 $OUT/log1.txt
