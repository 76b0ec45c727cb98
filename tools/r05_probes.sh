#!/bin/bash
# usage (GPU box): tools/r05_probes.sh <first|trsfwd|ablate|suite> -> gpurun_out/r05/: the opening measurements of round 5 (bounds of the judge's items),
# the forward-TRS probe, the boundary ablation and the full -m gpu suite + c3 timeline, as they were run (profiles/r05/plan.txt names what each produced).
# The libraries under ab_libs/ are variant builds: make -C road_segmentation_unet_amd/csrc VARIANT=<name> [DEV=1] EXTRA="-D...".
case "$1" in
first)
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
;;
trsfwd)
# GPU box: forward epilogue with the transposed (quad-merged) stores (PP_TRS_FWD=1) against the product build: fixed-shape layer times
# (alternating) and barrier-interval stamps
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/r05; mkdir -p $OUT; cd $REPO
SPECS="570,64,64,1 570,64,64,5 282,128,128,0 282,128,128,4 138,256,256,0 392,128,64,1 390,64,64,5 198,128,128,0 66,512,512,0"
for rep in 1 2 3; do
  for tag in base trsfwd; do
    lib=road_segmentation_unet_amd/librsu_hip.so; [ $tag = trsfwd ] && lib=ab_libs/librsu_trsfwd.so
    echo "$tag: $(RSU_LIB_PATH=$REPO/$lib timeout 300 python3 tools/pp_fixed.py $SPECS 2>/dev/null | tail -1)"
  done
done > $OUT/trsfwd_fixed.txt
cat $OUT/trsfwd_fixed.txt
for spec in "282 128 128 fwd 4 0" "570 64 64 fwd 4 1"; do
  echo "== $spec"; RSU_LIB_PATH=$REPO/ab_libs/librsu_dev_trsfwd.so timeout 300 python3 tools/pp_stamps_raw.py $spec 30 2>&1 | grep -v amdgpu.ids
done > $OUT/stamps_trsfwd.txt
grep -A45 "block 0" $OUT/stamps_trsfwd.txt | head -100
;;
ablate)
# GPU box: what the tile-boundary interval of igemm_pp (forward, transposed stores) is made of: stamps with timing ablations
# (RSU_FWD_DBG bits: 512 no stores, 8 no epilogue, 16 no bias initialisation, 64 no tile change, 256 no prefetch-stream tile change)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/r05; mkdir -p $OUT; cd $REPO
for dbg in 0 512 8 16 64 256 320; do
  for op in fwd bwd; do
  echo "== 282 128 128 $op cfg0 dbg $dbg"; RSU_STAMP_DBG=$dbg RSU_LIB_PATH=$REPO/ab_libs/librsu_dev_trsfwd.so timeout 300 python3 tools/pp_stamps_raw.py 282 128 128 $op 4 0 28 2>&1 | grep -v amdgpu.ids | head -36
  done
done > $OUT/stamps_ablate.txt
grep -c . $OUT/stamps_ablate.txt
;;
suite)
# GPU box: the whole -m gpu suite (parity figures -> gpurun_out/r05/parity_record.json), then the c3 step's timeline and bench lines
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/r05; mkdir -p $OUT; cd $REPO
rm -f $OUT/parity_record.json; export RSU_PARITY_RECORD=$OUT/parity_record.json
timeout 2400 python3 -m pytest tests -q -x -m gpu 2>&1 | grep -v amdgpu.ids | tail -8 > $OUT/gputests.txt; cat $OUT/gputests.txt
unset RSU_PARITY_RECORD
python3 bench.py --workload c3 --no_cpu_baseline > $OUT/bench_c3_start.json 2>/dev/null; cut -c1-200 $OUT/bench_c3_start.json
EXTRA_BENCH="--workload c3" ROUND=r05 bash tools/timeline.sh c3 > /dev/null 2>&1
head -5 $OUT/timeline_c3.txt
;;
*) echo "usage: $0 first|trsfwd|ablate|suite"; exit 2 ;;
esac
