#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/r04; mkdir -p $OUT; cd $REPO
timeout 900 python3 -m pytest tests/test_gpu_soak.py tests/test_gpu_net.py -x -q -m gpu -k "two_stream or bit_identical or small or dropout" 2>&1 | tail -4
bash tools/r04_abenv.sh wg_pair "RSU_WG_PAIR=0" "X=1" 3
bash tools/r04_abenv.sh wg_pair_c3 "RSU_WG_PAIR=0" "X=1" 2 "--workload c3"
