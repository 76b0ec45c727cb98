#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/r04; mkdir -p $OUT; cd $REPO
for rep in 1 2; do for sp in 128,128 120,136 136,120 112,144 144,112; do
RSU_SPLIT_CHIP=$sp timeout 600 python3 bench.py --steps 30 --warmup 5 --no_cpu_baseline --sustain_seconds 0 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('split $sp: %.1f patches/s | frac %.4f | ' % (d['value'], r['frac']) + ' '.join('%s %.0f/%.3f' % (k[8:], v['tflops'], v['wall_ms_per_step']) for k, v in r['by_kernel'].items()))
"
done; done | tee $OUT/split_sweep.txt
