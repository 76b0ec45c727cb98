#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/r04; mkdir -p $OUT; cd $REPO
timeout 1500 python3 -m pytest tests/test_gpu_cfg_matrix.py tests/test_gpu_ops.py tests/test_gpu_soak.py -x -q -m gpu 2>&1 | tail -8 > $OUT/check2_tests.txt
cat $OUT/check2_tests.txt
bash tools/r04_ab.sh boundary build_ab/librsu_hip_base.so road_segmentation_unet_amd/librsu_hip.so 3
