#!/bin/bash
# usage (GPU box): tools/split_sweep.sh name "spec1 spec2 ..." [reps] [extra bench args] -> gpurun_out/${ROUND:-r06}/split_<name>.txt: bench.py `value` for several
# RSU_SPLIT_CHIP settings (CUs the main / side stream plan for in the backward pass), alternating runs on one box; the update stays behind the pass
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/${ROUND:-r06}; mkdir -p $OUT
REPS=${3:-2}; EXTRA=${4:-}
for rep in $(seq 1 $REPS); do
  for spec in $2; do
    RSU_SPLIT_CHIP=$spec timeout 600 python3 $REPO/bench.py --steps 30 --warmup 5 --no_cpu_baseline --sustain_seconds 0 $EXTRA 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('split $spec: %.1f patches/s  %.3f ms/step | ' % (d['value'], d['ms_per_step']) + ' '.join('%s %.0f/%.3f' % (k[8:], v['tflops'], v['wall_ms_per_step']) for k, v in r['by_kernel'].items()))
"
  done
done | tee $OUT/split_$1.txt
