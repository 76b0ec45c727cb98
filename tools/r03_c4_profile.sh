REPO=$(pwd); mkdir -p $REPO/gpurun_out/r03; cd /tmp; export TMPDIR=/tmp RSU_WGRAD_STREAM=0
rm -rf /tmp/c4p; rocprofv3 --kernel-trace --stats -d /tmp/c4p -o k -- python3 $REPO/bench.py --workload c4 --steps 6 --warmup 2 --no_cpu_baseline --sustain_seconds 0 > /tmp/c4.log 2>&1
python3 - <<PY
import sqlite3, glob
db = sqlite3.connect(glob.glob("/tmp/c4p/**/*.db", recursive=True)[0]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]; ks = [t for t in tabs if "kernel_symbol" in t][0]
rows = cur.execute(f"select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start) from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc").fetchall()
ns = max([c for n, c, *_ in rows if "k_color_adjust" in n and "bwd" not in n] + [1])
tot = sum(r[2] for r in rows)
print("steps in trace:", ns, " total ms/step %.3f" % (tot / ns / 1e6))
for n, c, t, a in rows[:28]:
    print("%-70s %6.1f /step %8.3f ms/step avg %7.1f us" % (n.split("(")[0].replace("void ", "")[:70], c / ns, t / ns / 1e6, a / 1e3))
PY
