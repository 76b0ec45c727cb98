#!/bin/bash
# GPU box: timing ablations of igemm_pp on the layers of the c2 step (DEV library, make DEV=1 OUT=../librsu_hip_dev.so)
mkdir -p gpurun_out/r04
export RSU_LIB_PATH=$GRAFT_REPO_ROOT/road_segmentation_unet_amd/librsu_hip_dev.so
{
python3 tools/pp_subtract.py 282 128 128 fwd 4 0
python3 tools/pp_subtract.py 282 128 128 bwd 4 0
python3 tools/pp_subtract.py 66 512 512 fwd 4 0
python3 tools/pp_subtract.py 570 64 64 fwd 4 1
} > gpurun_out/r04/subtract.txt 2>&1
tail -70 gpurun_out/r04/subtract.txt
