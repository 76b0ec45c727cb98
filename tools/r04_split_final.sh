#!/bin/bash
# GPU box: bench.py value by the CU shares of the two backward streams (RSU_SPLIT_CHIP = main,side), alternating, final build of the round
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/r04; mkdir -p $OUT; cd $REPO
for rep in 1 2; do for sp in 128,128 144,112 160,96 136,120 112,144; do
  RSU_SPLIT_CHIP=$sp timeout 600 python3 bench.py --steps 30 --warmup 5 --no_cpu_baseline --sustain_seconds 0 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']['by_kernel']
print('RSU_SPLIT_CHIP=$sp: %.1f patches/s | ' % d['value'] + ' '.join('%s %.0f/%.3f' % (k[8:], v['tflops'], v['wall_ms_per_step']) for k, v in r.items()))"
done; done | tee $OUT/split_final.txt
