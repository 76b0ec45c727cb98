#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/r04; mkdir -p $OUT; cd $REPO
timeout 600 python3 tools/bench_layers.py 2>&1 | grep -v amdgpu.ids > $OUT/layers.txt
timeout 600 python3 tools/bench_layers.py --L 6 --dilated --B 1 2>&1 | grep -v amdgpu.ids > $OUT/layers_c3.txt
timeout 600 python3 tools/bench_layers.py --L 6 --dilated --B 1 --ncu 128 --ops bwd,wg 2>&1 | grep -v amdgpu.ids > $OUT/layers_c3_ncu128.txt
tail -1 $OUT/layers.txt; cat $OUT/layers_c3.txt; cat $OUT/layers_c3_ncu128.txt | tail -33
