#!/bin/bash
# GPU box: halo padding pixels not fetched (igemm_pp setup_a, igemm_wgpp issue_piece) against the build before: weight-gradient layer table, then the step
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/r04; mkdir -p $OUT; cd $REPO
for rep in 1 2; do for lib in librsu_hip_prev.so librsu_hip.so; do
  echo "== $lib"; RSU_LIB_PATH=$REPO/road_segmentation_unet_amd/$lib timeout 600 python3 tools/bench_layers.py --ops wg 2>&1 | grep -v amdgpu.ids | tail -1
done; done | tee $OUT/clip_wg_layers.txt
bash tools/r04_ab.sh clip road_segmentation_unet_amd/librsu_hip_prev.so road_segmentation_unet_amd/librsu_hip.so 4
