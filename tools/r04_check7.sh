#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/r04; mkdir -p $OUT; cd $REPO
run() { tag=$1; wl=$2; shift; shift; env "$@" timeout 600 python3 bench.py --workload $wl --steps 30 --warmup 5 --no_cpu_baseline --sustain_seconds 0 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('$wl $tag: %.1f patches/s | frac %.4f | ' % (d['value'], r['frac']) + ' '.join('%s %.0f/%.3f' % (k[8:], v['tflops'], v['wall_ms_per_step']) for k, v in r['by_kernel'].items()))
"; }
for rep in 1 2; do
for wl in c3 c2; do
run default $wl X=1
run group1 $wl RSU_WG_GROUP=1
run group2 $wl RSU_WG_GROUP=2
run group1_events $wl RSU_WG_GROUP=1 RSU_WG_EVENTS=1
done; done 2>&1 | tee $OUT/wg_group_noevents.txt
