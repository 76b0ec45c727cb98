#!/bin/bash
# usage (GPU box): tools/abenv.sh name "ENV_A" "ENV_B" [reps] [extra bench args] -> gpurun_out/${ROUND:-r06}/abenv_<name>.txt: bench.py `value` under two
# environment settings (space-separated VAR=value lists; "X=1" for none), alternating runs on one box
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/${ROUND:-r06}; mkdir -p $OUT
REPS=${4:-3}; EXTRA=${5:-}
for rep in $(seq 1 $REPS); do
  for tag in A B; do
    e=$2; [ $tag = B ] && e=$3
    env $e timeout 600 python3 $REPO/bench.py --steps 30 --warmup 5 --no_cpu_baseline --sustain_seconds 0 $EXTRA 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('$tag [$e]: %.1f patches/s | frac %.4f whole %.4f serial %.4f grouped %.4f | ' % (d['value'], r['frac'], r['whole_step_frac'], r['frac_serial_per_layer'], r['frac_single_stream_grouped']) + ' '.join('%s %.0f/%.3f' % (k[8:], v['tflops'], v['wall_ms_per_step']) for k, v in r['by_kernel'].items()))
"
  done
done | tee $OUT/abenv_$1.txt
