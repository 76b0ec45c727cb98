#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/r04; mkdir -p $OUT; cd $REPO
run() { tag=$1; shift; env "$@" timeout 600 python3 bench.py --workload c3 --steps 30 --warmup 5 --no_cpu_baseline --sustain_seconds 0 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('$tag: %.1f patches/s | frac %.4f | ' % (d['value'], r['frac']) + ' '.join('%s %.0f/%.3f' % (k[8:], v['tflops'], v['wall_ms_per_step']) for k, v in r['by_kernel'].items()))
"; }
for rep in 1 2; do
run default X=1
run wg_group_1 RSU_WG_GROUP=1
run wg_group_2 RSU_WG_GROUP=2
run wg_group_3 RSU_WG_GROUP=3
run one_stream_all RSU_WGRAD_STREAM=0
run split_160_96 RSU_SPLIT_CHIP=160,96
run split_96_160 RSU_SPLIT_CHIP=96,160
done 2>&1 | tee $OUT/c3_schedules.txt
bash tools/r04_ab.sh splitk_c2 build_ab/librsu_hip_base.so road_segmentation_unet_amd/librsu_hip.so 3
