#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/r04; mkdir -p $OUT; cd $REPO
timeout 1500 python3 -m pytest tests/test_gpu_net.py tests/test_gpu_soak.py tests/test_gpu_model.py -x -q -m gpu 2>&1 | tail -5
RSU_FWD_HALVES=1 timeout 900 python3 -m pytest tests/test_gpu_net.py -x -q -m gpu -k "c2_full_size or batch_of_four" 2>&1 | tail -3
for rep in 1 2 3; do for hv in 0 1 2; do
RSU_FWD_HALVES=$hv timeout 600 python3 bench.py --steps 30 --warmup 5 --no_cpu_baseline --sustain_seconds 0 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('halves $hv: %.1f patches/s | frac %.4f whole %.4f | ' % (d['value'], r['frac'], r['whole_step_frac']) + ' '.join('%s %.0f/%.3f' % (k[8:], v['tflops'], v['wall_ms_per_step']) for k, v in r['by_kernel'].items()))
"
done; done | tee $OUT/fwd_halves.txt
