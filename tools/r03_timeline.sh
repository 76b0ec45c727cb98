#!/bin/bash
# usage (GPU box): tools/r03_timeline.sh "<RSU_WG_GROUP values>" -> gpurun_out/r03d/timeline_g<G>.txt: start / end / queue of every kernel of
# the last steady-state step of the default two-stream schedule (rocprofv3 --kernel-trace)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r03d; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for G in $1; do
  export RSU_WG_GROUP=$G
  rm -rf $OUT/prof_$G
  rocprofv3 --kernel-trace -d $OUT/prof_$G -o t -- python3 $REPO/bench.py --steps 6 --warmup 2 --no_cpu_baseline --sustain_seconds 0 > $OUT/prof_$G.log 2>&1
  python3 - <<PY
import sqlite3, glob
dbs = glob.glob("$OUT/prof_$G/**/*.db", recursive=True)
db = sqlite3.connect(dbs[0]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]; ks = [t for t in tabs if "kernel_symbol" in t][0]
cols = [r[1] for r in cur.execute(f"pragma table_info({kd})")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else cols[0])
rows = cur.execute(f"select s.kernel_name, d.start, d.end, d.{qcol} from {kd} d join {ks} s on d.kernel_id=s.id order by d.start").fetchall()
# steps are delimited by k_momentum; take the step in front of the timed region's end: the 6th from the end is inside the timed loop
mom = [i for i, r in enumerate(rows) if "k_color_adjust" in r[0] and "bwd" not in r[0]]   # a step begins with the colour adjust of its forward pass
# the timed region holds 6 steps; the instrumented pass behind it holds 1 + 3 more: pick the 3rd timed step
i0, i1 = mom[-8], mom[-7]
seg = rows[i0:i1]
t0 = seg[0][1]
def short(n):
    for k in ("igemm_wg_group", "igemm_wgpp", "igemm_wgp64", "igemm_wgt", "igemm_wg1", "igemm_wgrad", "k_conv_first_fwd", "k_update_pack", "igemm_pp", "igemm_fwd2", "igemm_ct", "k_reduce_slabs_many", "k_reduce_slabs", "k_pool_skip", "k_maxpool", "k_head", "k_momentum", "k_pack", "k_color_adjust_bwd", "k_color_adjust", "k_scatter"):
        if k in n: return k
    return n[:30]
with open("$OUT/timeline_g$G.txt", "w") as f:
    f.write("step of %.3f ms (columns: start us, end us, duration us, queue, kernel)\n" % ((seg[-1][2] - t0) / 1e6))
    for n, s, e, q in seg:
        f.write("%9.1f %9.1f %8.1f  q%-4s %s\n" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, short(n)))
print(open("$OUT/timeline_g$G.txt").read())
PY
  rm -rf $OUT/prof_$G
done
