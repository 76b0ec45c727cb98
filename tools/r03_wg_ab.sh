#!/bin/bash
# usage (GPU box): tools/r03_wg_ab.sh -> gpurun_out/r03b/: the grouped weight-gradient launches (RSU_WG_GROUP) against per-layer launches
OUT=gpurun_out/r03b; mkdir -p $OUT
if [ -z "$NOTESTS" ]; then
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "wgrad_group or bwd_weight or pingpong_wgrad" > $OUT/t_ops.log 2>&1; tail -3 $OUT/t_ops.log
timeout 900 python -m pytest tests/test_gpu_net.py tests/test_gpu_soak.py tests/test_gpu_dp.py -x -q > $OUT/t_net.log 2>&1; tail -3 $OUT/t_net.log
fi
for G in ${SPECS:-0 1 5,4 4,5 3,3,3 2,3,4 all}; do
  RSU_WG_GROUP=$G python bench.py --no_cpu_baseline --sustain_seconds 0 > $OUT/bench_g$G.json 2> $OUT/bench_g$G.err
  python - <<PY
import json
d=json.load(open("$OUT/bench_g$G.json"))
r=d["roofline"]
print("G=$G value %.1f ms %.3f frac %.4f conv_ms %.3f" % (d["value"], d["ms_per_step"], r["frac"], r["conv_ms_per_step"]), {k:(round(v["ms_per_step"],3),v["launches"]) for k,v in r["by_kernel"].items()})
PY
done
