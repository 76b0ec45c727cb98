#!/bin/bash
# usage (GPU box): tools/r03_predict_profile.sh -> gpurun_out/r03/predict_c5_kernels.txt: kernel time by family of ConvolutionalModel.predict on one
# 604x604 image (config 5: L=6 dilated, stride 12, 6-way ensemble), second call (rocprofv3 --kernel-trace --stats)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $REPO/gpurun_out/r03; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pp5; rocprofv3 --kernel-trace --stats -d /tmp/pp5 -o k -- python3 $REPO/tools/bench_predict.py --L 6 --dilated --images 1 --stride 12 --size 604 --batch 8 > /tmp/pp5.log 2>&1
tail -1 /tmp/pp5.log
python3 - <<PY
import sqlite3, glob, collections
db = sqlite3.connect(glob.glob("/tmp/pp5/**/*.db", recursive=True)[0]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]; ks = [t for t in tabs if "kernel_symbol" in t][0]
rows = cur.execute(f"select s.kernel_name, count(*), sum(d.end-d.start) from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc").fetchall()
fam = collections.OrderedDict()
for n, c, t in rows:
    k = n.split("(")[0].replace("void ", "")
    for f in ("igemm_pp", "igemm_fwd2", "igemm_ct", "k_conv_first_fwd", "k_maxpool", "k_color_adjust", "k_head", "k_extract", "k_overlap", "k_pack", "at::native", "rocclr"):
        if f in k: k = f; break
    e = fam.setdefault(k[:50], [0, 0]); e[0] += c; e[1] += t
tot = sum(v[1] for v in fam.values())
print("kernel time in the trace (both predict calls + tuning): %.1f ms" % (tot / 1e6))
for k, (c, t) in sorted(fam.items(), key=lambda kv: -kv[1][1])[:16]:
    print("%-40s launches %6d  %9.2f ms  %5.1f %%" % (k, c, t / 1e6, 100.0 * t / tot))
PY
