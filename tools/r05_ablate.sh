#!/bin/bash
# GPU box: what the tile-boundary interval of igemm_pp (forward, transposed stores) is made of: stamps with timing ablations
# (RSU_FWD_DBG bits: 512 no stores, 8 no epilogue, 16 no bias initialisation, 64 no tile change, 256 no prefetch-stream tile change)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/r05; mkdir -p $OUT; cd $REPO
for dbg in 0 512 8 16 64 256 320; do
  for op in fwd bwd; do
  echo "== 282 128 128 $op cfg0 dbg $dbg"; RSU_STAMP_DBG=$dbg RSU_LIB_PATH=$REPO/ab_libs/librsu_dev_trsfwd.so timeout 300 python3 tools/pp_stamps_raw.py 282 128 128 $op 4 0 28 2>&1 | grep -v amdgpu.ids | head -36
  done
done > $OUT/stamps_ablate.txt
grep -c . $OUT/stamps_ablate.txt
