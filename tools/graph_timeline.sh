#!/bin/bash
# usage (GPU box): tools/graph_timeline.sh -> gpurun_out/<ROUND>/graph_timeline.txt: rocprofv3 --kernel-trace of tools/graph_probe.py; one eager step and one
# hipGraph replay of the same c2 training step side by side: queues used, busy time and idle gaps per queue, kernel durations
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/${ROUND:-r06}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof_graph
rocprofv3 --kernel-trace -d $OUT/prof_graph -o t -- python3 $REPO/tools/graph_probe.py > $OUT/graph_probe.log 2>&1
python3 - <<PY
import sqlite3, glob, collections
dbs = glob.glob("$OUT/prof_graph/**/*.db", recursive=True)
db = sqlite3.connect(dbs[0]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]; ks = [t for t in tabs if "kernel_symbol" in t][0]
cols = [r[1] for r in cur.execute(f"pragma table_info({kd})")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else cols[0])
rows = cur.execute(f"select s.kernel_name, d.start, d.end, d.{qcol} from {kd} d join {ks} s on d.kernel_id=s.id order by d.start").fetchall()
first = [i for i, r in enumerate(rows) if "k_color_adjust" in r[0] and "bwd" not in r[0]]
def describe(tag, i0, i1):
    seg = rows[i0:i1]
    t0 = seg[0][1]
    q = collections.OrderedDict()
    for n, s, e, qq in seg:
        a = q.setdefault(qq, [0, 0.0, 0.0, None, 0])
        if a[3] is not None:
            gap = (s - a[3]) / 1e3
            if gap > 0.3:
                a[2] += gap; a[4] += 1
        a[0] += 1; a[1] += (e - s) / 1e3; a[3] = e
    out = ["%s: step of %.3f ms, %d kernels on %d queues, kernel time summed %.1f us" % (tag, (seg[-1][2] - t0) / 1e6, len(seg), len(q), sum(a[1] for a in q.values()))]
    for qq, a in q.items():
        out.append("   queue %s: %3d kernels, busy %7.1f us, %3d idle gaps > 0.3 us summing %6.1f us (%.1f us each)" % (qq, a[0], a[1], a[4], a[2], a[2] / max(1, a[4])))
    return out
# order of the run: eager net: tune + 5 + 40 steps; graph net: tune + 5 eager steps, then 40 replays
lines = describe("eager step", first[25], first[26]) + describe("hipGraph replay", first[-12], first[-11])
lines.append(open("$OUT/graph_probe.log").read().strip().splitlines()[-2])
open("$OUT/graph_timeline.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
rm -rf $OUT/prof_graph
