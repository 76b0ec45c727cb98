#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/r04; mkdir -p $OUT; cd $REPO
run() { tag=$1; shift; env "$@" timeout 600 python3 bench.py --workload c3 --steps 30 --warmup 5 --no_cpu_baseline --sustain_seconds 0 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('$tag: %.1f patches/s | frac %.4f | ' % (d['value'], r['frac']) + ' '.join('%s %.0f/%.3f' % (k[8:], v['tflops'], v['wall_ms_per_step']) for k, v in r['by_kernel'].items()))
"; }
run default X=1
for g in 2 3 4; do for sp in 144,112 160,96 176,80 192,64; do
run "group$g split$sp" RSU_WG_GROUP=$g RSU_SPLIT_CHIP=$sp
done; done 2>&1 | tee $OUT/c3_schedules2.txt
run default X=1 | tee -a $OUT/c3_schedules2.txt
