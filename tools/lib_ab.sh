#!/bin/bash
# usage (GPU box): tools/lib_ab.sh name libA.so libB.so -> gpurun_out/r03/lib_ab_<name>.txt: bench.py `value` (default two-stream schedule) and the
# single-stream value under two builds of the library, alternating runs on one box
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/r03; mkdir -p $OUT
for rep in 1 2 3; do
  for tag in A B; do
    lib=$2; [ $tag = B ] && lib=$3
    v2=$(RSU_LIB_PATH=$REPO/$lib python3 $REPO/bench.py --steps 30 --warmup 5 --no_cpu_baseline --sustain_seconds 0 2>/dev/null | grep -o '"value": [0-9.]*' | head -1)
    v1=$(RSU_WGRAD_STREAM=0 RSU_LIB_PATH=$REPO/$lib python3 $REPO/bench.py --steps 30 --warmup 5 --no_cpu_baseline --sustain_seconds 0 2>/dev/null | grep -o '"value": [0-9.]*' | head -1)
    echo "$tag $lib: two streams $v2 | one stream $v1"
  done
done | tee $OUT/lib_ab_$1.txt
