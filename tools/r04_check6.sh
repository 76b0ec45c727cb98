#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/r04; mkdir -p $OUT; cd $REPO
bash tools/r04_abenv.sh c4_split "RSU_KSPLIT=0 RSU_PP_DIL2=0" "X=1" 2 "--workload c4"
for rep in 1 2; do
  RSU_KSPLIT=0 RSU_PP_DIL2=0 timeout 600 python3 tools/bench_predict.py --L 6 --dilated 2>&1 | tail -1 | sed 's/^/A [RSU_KSPLIT=0 RSU_PP_DIL2=0]: /'
  timeout 600 python3 tools/bench_predict.py --L 6 --dilated 2>&1 | tail -1 | sed 's/^/B [default]: /'
done | tee $OUT/predict_c5_ab.txt
