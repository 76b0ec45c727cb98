#!/bin/bash
# GPU box, round 5 opening measurements -> gpurun_out/r05/: (1) the benchmarked batch against the oracle (new assertions), (2) the upper bound of a
# 1-bit ReLU mask (backward-data with and without its mask source, tuned shapes, whole chip and half chip), (3) tile plans at 256 and 128 CUs,
# (4) per-image split-K probe A/B, (5) barrier-interval stamps of igemm_pp (developer build)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/r05; mkdir -p $OUT; cd $REPO
export RSU_PARITY_RECORD=$OUT/parity_record.json
timeout 1500 python3 -m pytest tests/test_gpu_net.py -q -x -m gpu -k "c2_full_size_gradients or benchmarked_batch or step_properties" -s 2>&1 | grep -v amdgpu.ids | tail -15 > $OUT/parity_tests.txt
cat $OUT/parity_tests.txt
for ncu in 0 128; do
  echo "== ncu $ncu"; timeout 600 python3 tools/bench_layers.py --ops bwd,bwdnm --ncu $ncu 2>&1 | grep -v amdgpu.ids
done > $OUT/mask_bound.txt
cat $OUT/mask_bound.txt
RSU_PLAN_DEBUG=1 timeout 600 python3 bench.py --steps 2 --warmup 1 --no_cpu_baseline --sustain_seconds 0 2>&1 >/dev/null | grep "plan fwd2" | sort | uniq -c > $OUT/tile_util_step.txt
wc -l $OUT/tile_util_step.txt
bash tools/r04_abenv.sh perimg_c2 "RSU_KSPLIT_PERIMG=0" "RSU_KSPLIT_PERIMG=1" 2; mv gpurun_out/r04/abenv_perimg_c2.txt $OUT/
bash tools/r04_abenv.sh perimg_c4 "RSU_KSPLIT_PERIMG=0" "RSU_KSPLIT_PERIMG=1" 2 "--workload c4"; mv gpurun_out/r04/abenv_perimg_c4.txt $OUT/
for spec in "282 128 128 fwd 4 0" "282 128 128 bwd 4 0" "570 64 64 fwd 4 1" "570 64 64 bwd 4 1" "138 256 256 fwd 4 0"; do
  echo "== $spec"; RSU_LIB_PATH=$REPO/ab_libs/librsu_dev_r04.so timeout 300 python3 tools/pp_stamps_raw.py $spec 60 2>&1 | grep -v amdgpu.ids
done > $OUT/stamps.txt
head -30 $OUT/stamps.txt
