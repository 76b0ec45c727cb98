#!/bin/bash
# usage (GPU box): [SPECS="0 1 2 ..."] tools/r03_wg_ab.sh -> gpurun_out/r03b/wg_group_schedules.txt: the two-stream step under different groupings of
# the weight gradients (RSU_WG_GROUP = blocks per group; 0 = one launch per layer) and the single-stream schedule, same box, alternating
OUT=gpurun_out/r03b; mkdir -p $OUT
TAB=$OUT/wg_group_schedules.txt
echo "bench.py --no_cpu_baseline --sustain_seconds 0 under RSU_WG_GROUP (two-stream timed region; the serialised roofline pass uses the same grouping), one box:" > $TAB
for rep in 1 2; do
for G in ${SPECS:-0 1 2 3 3,3,3 5,4 all}; do
  RSU_WG_GROUP=$G python bench.py --no_cpu_baseline --sustain_seconds 0 > $OUT/bench_g$G.json 2> $OUT/bench_g$G.err
  python - >> $TAB <<PY
import json
d=json.load(open("$OUT/bench_g$G.json"))
r=d["roofline"]; w=r["by_kernel"]["conv3x3_bwd_weight"]
print("RSU_WG_GROUP=%-6s %7.1f patches/s  %.3f ms/step | serialised pass: weight gradients %.3f ms in %2d launches, roofline.frac %.4f" % ("$G", d["value"], d["ms_per_step"], w["ms_per_step"], w["launches"], r["frac"]))
PY
done
RSU_WGRAD_STREAM=0 python bench.py --no_cpu_baseline --sustain_seconds 0 > $OUT/bench_1stream.json 2> $OUT/bench_1stream.err
python - >> $TAB <<PY
import json
d=json.load(open("$OUT/bench_1stream.json"))
r=d["roofline"]; w=r["by_kernel"]["conv3x3_bwd_weight"]
print("one stream (grouped)  %7.1f patches/s  %.3f ms/step | serialised pass: weight gradients %.3f ms in %2d launches, roofline.frac %.4f" % (d["value"], d["ms_per_step"], w["ms_per_step"], w["launches"], r["frac"]))
PY
done
cat $TAB
