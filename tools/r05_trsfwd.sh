#!/bin/bash
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
