#!/bin/bash
# usage: tools/gpu_profile_cmd.sh <tag> <script.py> [args...]  (on the GPU box) -> gpurun_out/<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats)
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
S=$REPO/$1; shift
mkdir -p $REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_$TAG -o $TAG -- python3 $S "$@" > $REPO/gpurun_out/prof_$TAG.log 2>&1
python3 - <<PY
import sqlite3, glob, csv
db = sqlite3.connect(glob.glob("$REPO/gpurun_out/prof_$TAG/*.db")[0])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
rows = cur.execute(f"select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start) from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
with open("$REPO/gpurun_out/${TAG}_kernel_stats.csv", "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for n, c, t, a, mn, mx in rows:
        w.writerow([n, c, t, "%.1f" % a, "%.2f" % (100.0 * t / tot), mn, mx])
print("kernels:", len(rows), "total ms:", tot / 1e6)
PY
tail -1 $REPO/gpurun_out/prof_$TAG.log | cut -c1-250
