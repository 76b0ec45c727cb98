#!/bin/bash
# usage (GPU box): tools/r04_fixed_ab.sh libA.so libB.so [libC.so ...]: tools/pp_fixed.py on the big layers of c2 under each build, three rounds alternating
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $REPO/gpurun_out/r04; cd $REPO
for rep in 1 2 3; do for lib in "$@"; do
  echo -n "$(basename $lib): "; RSU_LIB_PATH=$REPO/$lib python3 tools/pp_fixed.py 570,64,64,1 282,128,128,0 392,128,64,1 138,256,256,0 2>/dev/null
done; done | tee $REPO/gpurun_out/r04/fixed_ab.txt
