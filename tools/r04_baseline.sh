#!/bin/bash
# round-4 baseline on one box (before any kernel change): planner utilisation of every conv launch of c2, per-layer tables of c2 and c3,
# the c5 inference time under round 2's and round 3's trees (alternating), and the headline bench line
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/r04; mkdir -p $OUT
cd $REPO
RSU_PLAN_DEBUG=1 RSU_WGRAD_STREAM=0 timeout 600 python3 bench.py --steps 3 --warmup 1 --no_cpu_baseline --sustain_seconds 0 2>&1 | grep -E "plan fwd2|\"value\"" | sort | uniq -c | sort -k4,4 -k5,5n > $OUT/tile_util_before.txt
timeout 600 python3 tools/bench_layers.py > $OUT/layers_before.txt 2>&1
timeout 600 python3 tools/bench_layers.py --L 6 --dilated --B 1 > $OUT/layers_c3_before.txt 2>&1
for rep in 1 2; do
  (cd $REPO/build_ab/r02 && timeout 600 python3 tools/bench_predict.py --L 6 --dilated 2>&1 | tail -1 | sed 's/^/r02 tree: /') >> $OUT/predict_ab.txt
  timeout 600 python3 tools/bench_predict.py --L 6 --dilated 2>&1 | tail -1 | sed 's/^/r03 tree: /' >> $OUT/predict_ab.txt
done
timeout 900 python3 bench.py --no_cpu_baseline > $OUT/bench_before.json 2> $OUT/bench_before.err
tail -3 $OUT/predict_ab.txt; tail -2 $OUT/layers_before.txt; tail -2 $OUT/layers_c3_before.txt; cat $OUT/bench_before.json | head -c 600
