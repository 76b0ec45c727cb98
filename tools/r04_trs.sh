#!/bin/bash
# GPU box: epilogue stores transposed across the lanes (igemm_pp TRS) vs the build before (librsu_hip_base.so): operator tests, then the
# c2 layer table under both builds (alternating), then the step
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/r04; mkdir -p $OUT; cd $REPO
timeout 900 python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_cfg_matrix.py -q -x -m gpu 2>&1 | tail -5 > $OUT/trs_tests.txt
cat $OUT/trs_tests.txt
for rep in 1 2; do
  for tag in base new; do
    lib=road_segmentation_unet_amd/librsu_hip.so; [ $tag = base ] && lib=road_segmentation_unet_amd/librsu_hip_base.so
    echo "== $tag (rep $rep)"; RSU_LIB_PATH=$REPO/$lib timeout 600 python3 tools/bench_layers.py --ops fwd,bwd 2>&1 | grep -v amdgpu.ids
  done
done > $OUT/trs_layers.txt
cat $OUT/trs_layers.txt
bash tools/r04_ab.sh trs road_segmentation_unet_amd/librsu_hip_base.so road_segmentation_unet_amd/librsu_hip.so 3
