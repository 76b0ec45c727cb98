#!/bin/bash
# usage (GPU box): tools/r03_trace.sh "<G values>" -> gpurun_out/r03c/stats_g<G>.txt: per-kernel totals of a single-stream bench run
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r03c; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export RSU_WGRAD_STREAM=0
for G in $1; do
  export RSU_WG_GROUP=$G
  rm -rf $OUT/prof_$G
  rocprofv3 --kernel-trace --stats -d $OUT/prof_$G -o t -- python3 $REPO/bench.py --steps 10 --warmup 2 --no_cpu_baseline --sustain_seconds 0 > $OUT/prof_$G.log 2>&1
  python3 - <<PY
import sqlite3, glob
dbs = glob.glob("$OUT/prof_$G/**/*.db", recursive=True)
db = sqlite3.connect(dbs[0]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]; ks = [t for t in tabs if "kernel_symbol" in t][0]
rows = cur.execute(f"select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start) from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc").fetchall()
with open("$OUT/stats_g$G.txt","w") as f:
    for n,c,t,a in rows[:30]:
        f.write("%-90s calls %5d total_ms %9.3f avg_us %8.1f\n" % (n[:90], c, t/1e6, a/1e3))
print(open("$OUT/stats_g$G.txt").read())
PY
  rm -rf $OUT/prof_$G
done
