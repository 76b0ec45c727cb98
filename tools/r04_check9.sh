#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/r04; mkdir -p $OUT; cd $REPO
timeout 900 python3 -m pytest tests/test_gpu_net.py -x -q -m gpu -k "early_update or train_steps" 2>&1 | tail -4
run() { tag=$1; wl=$2; shift; shift; env "$@" timeout 600 python3 bench.py --workload $wl --steps 30 --warmup 5 --no_cpu_baseline --sustain_seconds 0 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('$wl $tag: %.1f patches/s (%.3f ms) | frac %.4f | ' % (d['value'], d['ms_per_step'], r['frac']) + ' '.join('%s %.0f/%.3f' % (k[8:], v['tflops'], v['wall_ms_per_step']) for k, v in r['by_kernel'].items()))
"; }
for rep in 1 2; do
run off c3 RSU_EARLY_UPDATE=0
run level4 c3 RSU_EARLY_UPDATE=4
run level3 c3 RSU_EARLY_UPDATE=3
run level2 c3 RSU_EARLY_UPDATE=2
run off c2 RSU_EARLY_UPDATE=0
run level3 c2 RSU_EARLY_UPDATE=3
run level2 c2 RSU_EARLY_UPDATE=2
run off c4 RSU_EARLY_UPDATE=0
run level3 c4 RSU_EARLY_UPDATE=3
done 2>&1 | tee $OUT/early_update.txt
