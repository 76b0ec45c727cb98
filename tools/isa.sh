#!/bin/bash
# usage: tools/isa.sh <file.hip (in csrc)> <kernel name substring>  -- compile with -save-temps and summarise the ISA
REPO=/root/repo; F=$1; PAT=$2
mkdir -p /tmp/st
( cd $REPO/road_segmentation_unet_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c $F -o /tmp/st/${F%.hip}.o -save-temps=obj 2>&1 | grep -E "error|warning" | head -20 )
S=/tmp/st/${F%.hip}-hip-amdgcn-amd-amdhsa-gfx950.s
grep -E "^\s+\.(vgpr_count|name:|vgpr_spill|private_segment_fixed)" $S | paste - - - - | awk '{print $2,"scratch="$4,"vgpr="$6,"spill="$8}' | sed 's/_Z1[0-9]igemm_//' | cut -c1-110
[ -n "$PAT" ] && python3 $REPO/tools/isa_summary.py $S "$PAT"
