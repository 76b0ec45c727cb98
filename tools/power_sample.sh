#!/bin/bash
# usage (GPU box): tools/power_sample.sh name [lib.so ...] -> gpurun_out/<ROUND>/power_<name>.txt: rocm-smi socket power / clocks sampled every ~0.3 s while
# bench.py's sustained loop runs (12 s), once per library build given (default: the product build)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/${ROUND:-r06}; mkdir -p $OUT; cd $REPO
NAME=$1; shift
LIBS=${@:-road_segmentation_unet_amd/librsu_hip.so}
for lib in $LIBS; do
  echo "== $lib"
  RSU_LIB_PATH=$REPO/$lib python3 bench.py --steps 20 --warmup 3 --no_cpu_baseline --sustain_seconds 12 > /tmp/pb.json 2>/dev/null &
  BP=$!
  sleep 7
  for i in $(seq 1 14); do
    rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | sed 's/.*: //' | tr '\n' ' '; echo
    sleep 0.3
  done
  wait $BP
  grep -o '"sustained": {[^}]*}' /tmp/pb.json
done | tee $OUT/power_$NAME.txt
