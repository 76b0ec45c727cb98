#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/r04; mkdir -p $OUT; cd $REPO
timeout 1500 python3 -m pytest tests/test_gpu_cfg_matrix.py tests/test_gpu_ops.py -x -q -m gpu -k "pingpong or dilated or split_k or conv2d_fwd or conv2d_bwd_data" 2>&1 | tail -6
bash tools/r04_abenv.sh c3_d2 "RSU_PP_DIL2=0" "X=1" 2 "--workload c3"
bash tools/r04_abenv.sh c2_split "RSU_KSPLIT=0 RSU_COB_GROUP=0" "X=1" 3
bash tools/r04_abenv.sh c2_cobgroup "RSU_COB_GROUP=0" "X=1" 2
