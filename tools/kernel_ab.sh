#!/bin/bash
# usage (GPU box): tools/kernel_ab.sh name libA.so libB.so [two-stream]  -> gpurun_out/r03/kernel_ab_<name>.txt
# per-kernel average durations of a short bench run (rocprofv3 --kernel-trace --stats) under two builds of the library, for the
# bandwidth-bound kernels; single-stream schedule unless "two-stream" is given
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r03; mkdir -p $OUT
NAME=$1; A=$2; B=$3
[ "$4" = "two-stream" ] || export RSU_WGRAD_STREAM=0
cd /tmp && export TMPDIR=/tmp
for tag in A B A B; do
  lib=$A; [ $tag = B ] && lib=$B
  export RSU_LIB_PATH=$REPO/$lib
  rm -rf /tmp/kab; rocprofv3 --kernel-trace --stats -d /tmp/kab -o k -- python3 $REPO/bench.py --steps 8 --warmup 2 --no_cpu_baseline --sustain_seconds 0 > /tmp/kab.log 2>&1
  echo "== $tag $lib: $(grep -o '"value": [0-9.]*' /tmp/kab.log | head -1) patches/s"
  python3 - <<PY
import sqlite3, glob
db = sqlite3.connect(glob.glob("/tmp/kab/**/*.db", recursive=True)[0]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]; ks = [t for t in tabs if "kernel_symbol" in t][0]
for n, c, a in cur.execute(f"select s.kernel_name, count(*), avg(d.end-d.start) from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc"):
    if ("igemm" in n and "igemm_fwd2" not in n and "igemm_wg1" not in n and "igemm_wgrad_kernel" not in n and "igemm_wgt" not in n) or "at::" in n or "rocclr" in n: continue
    print("   %-60s calls %4d  avg %7.1f us  total %8.1f us" % (n.split("(")[0][:60], c, a / 1e3, a * c / 1e3))
PY
done > $OUT/kernel_ab_$NAME.txt 2>&1
cat $OUT/kernel_ab_$NAME.txt
