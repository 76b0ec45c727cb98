#!/bin/bash
# usage (GPU box): tools/dp_budget.sh [reps] -> gpurun_out/${ROUND:-r06}/dp_budget.txt: the c2 and c4-share steps with the backward launches planned for
# 256 / 240 / 224 / 208 CUs (dist.EXCHANGE_CANDIDATES: the budgets a data-parallel run offers so that RCCL's channel workgroups find CUs), one GPU, no
# exchange, the update behind the pass as under an exchange (RSU_FUSED_WGRAD=0); alternating runs on one box. VERDICT r5 item 4.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/${ROUND:-r06}; mkdir -p $OUT
REPS=${1:-2}
for wl in c2 c4; do
  for rep in $(seq 1 $REPS); do
    for b in 256 240 224 208 192; do
      RSU_FUSED_WGRAD=0 RSU_BENCH_BWD_BUDGET=$b timeout 600 python3 $REPO/bench.py --workload $wl --steps 30 --warmup 5 --no_cpu_baseline --sustain_seconds 0 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('$wl budget $b: %.1f patches/s  %.3f ms/step | ' % (d['value'], d['ms_per_step']) + ' '.join('%s %.0f/%.3f' % (k[8:], v['tflops'], v['wall_ms_per_step']) for k, v in r['by_kernel'].items()))
"
    done
  done
done | tee $OUT/dp_budget.txt
