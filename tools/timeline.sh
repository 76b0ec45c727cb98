#!/bin/bash
# usage (GPU box): tools/timeline.sh <tag> [env assignments...] -> gpurun_out/${ROUND:-r06}/timeline_<tag>.txt: start / end / duration / queue of every kernel of
# one steady-state step of bench.py's timed loop (rocprofv3 --kernel-trace), with the idle gap in front of each kernel on its queue
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/${ROUND:-r06}; mkdir -p $OUT
TAG=$1; shift
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof_$TAG
rocprofv3 --kernel-trace -d $OUT/prof_$TAG -o t -- python3 $REPO/bench.py --steps 6 --warmup 2 --no_cpu_baseline --sustain_seconds 0 $EXTRA_BENCH > $OUT/prof_$TAG.log 2>&1
python3 - <<PY
import sqlite3, glob
dbs = glob.glob("$OUT/prof_$TAG/**/*.db", recursive=True)
db = sqlite3.connect(dbs[0]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]; ks = [t for t in tabs if "kernel_symbol" in t][0]
cols = [r[1] for r in cur.execute(f"pragma table_info({kd})")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else cols[0])
rows = cur.execute(f"select s.kernel_name, d.start, d.end, d.{qcol}, d.grid_size_x, d.workgroup_size_x from {kd} d join {ks} s on d.kernel_id=s.id order by d.start").fetchall()
first = [i for i, r in enumerate(rows) if "k_color_adjust" in r[0] and "bwd" not in r[0]]   # a step begins with the colour adjust of its forward pass
# bench.py: tune (1 fwd+bwd) + priming + 2 warm-up + 6 timed + 3 instrumented passes of 4 steps + tune: take the 4th timed step
steps = [i for i in first]
i0, i1 = steps[-16], steps[-15]
seg = rows[i0:i1]
t0 = seg[0][1]
def short(n):
    for k in ("igemm_wg_group", "igemm_wgpp", "igemm_wgp64", "igemm_wgt", "igemm_wg1", "igemm_wgrad", "k_conv_first_fwd", "k_update_pack", "igemm_pp", "igemm_fwd2", "igemm_ct", "k_reduce_slabs_many", "k_reduce_slabs", "k_pool_skip", "k_maxpool", "k_head", "k_momentum", "k_pack", "k_color_adjust_bwd", "k_color_adjust", "k_scatter", "k_pp_splitk"):
        if k in n: return k
    return n[:30]
last_end = {}
with open("$OUT/timeline_$TAG.txt", "w") as f:
    f.write("step of %.3f ms (columns: start us, end us, duration us, idle gap on its queue us, queue, workgroups, kernel)\n" % ((seg[-1][2] - t0) / 1e6))
    for n, s, e, q, gx, wx in seg:
        gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
        last_end[q] = e
        f.write("%9.1f %9.1f %8.1f %7.1f  q%-4s %5d %s\n" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, gap, q, gx // max(1, wx), short(n)))
print(open("$OUT/timeline_$TAG.txt").read()[:6000])
PY
rm -rf $OUT/prof_$TAG
