#!/bin/bash
# GPU box: forward / backward-data of the big c2 layers at a fixed tile shape by strip width (RSU_FWD2_LSW: 0 = the cost model's pick).
# Needs the temporary patch described in profiles/r04/lsw_sweep.txt (the product library has no such switch).
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $REPO/gpurun_out/r04; cd $REPO
for rep in 1 2; do for l in 0 3 4 5 6; do
  echo -n "lsw $l: "; RSU_FWD2_LSW=$l python3 tools/pp_fixed.py 570,64,64,1 282,128,128,0 392,128,64,1 138,256,256,0 198,128,128,0 102,256,256,0 66,512,512,0 2>/dev/null
done; done | tee $REPO/gpurun_out/r04/lsw_sweep.txt
