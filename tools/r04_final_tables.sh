#!/bin/bash
# GPU box: the round's closing tables: planner utilisation of every conv launch of c2 (as tile_util_before.txt), per-layer tables of c2 and c3
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/r04; mkdir -p $OUT; cd $REPO
RSU_PLAN_DEBUG=1 RSU_WGRAD_STREAM=0 timeout 600 python3 bench.py --steps 3 --warmup 1 --no_cpu_baseline --sustain_seconds 0 2>&1 | grep -E "plan fwd2|\"value\"" | cut -c1-260 | sort | uniq -c | sort -k4,4 -k5,5n > $OUT/tile_util_after.txt
timeout 600 python3 tools/bench_layers.py 2>&1 | grep -v amdgpu.ids > $OUT/layers.txt
timeout 600 python3 tools/bench_layers.py --L 6 --dilated --B 1 2>&1 | grep -v amdgpu.ids > $OUT/layers_c3.txt
tail -2 $OUT/layers.txt; tail -1 $OUT/layers_c3.txt; wc -l $OUT/tile_util_after.txt
