#!/bin/bash
# usage: tools/pmc_cmd.sh <tag> <script.py> [args...]  -> per-kernel averages of SQ counters for `python3 script.py args` (two passes)
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
S=$REPO/$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT -o p1 -- python3 $S "$@" > $OUT/log1.txt 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM --kernel-trace --output-format csv -d $OUT -o p2 -- python3 $S "$@" > $OUT/log2.txt 2>&1
python3 - <<PY
import csv, glob, collections
dur = collections.defaultdict(list)
for f in sorted(glob.glob("$OUT/p1_kernel_trace.csv")):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"][:60]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for f in sorted(glob.glob("$OUT/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        if "igemm" not in k: continue
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        print(k, "us=%.1f" % (sum(dur[k]) / max(1, len(dur[k]))), {c: round(sum(v)/len(v)) for c, v in d.items()})
PY
tail -2 $OUT/log1.txt
