mkdir -p gpurun_out/r03
for rep in 1 2; do
for cfg in "2:128,128" "3:128,128" "3:144,112" "3:160,96" "3:136,120" "2:160,96" "3:192,64"; do
  wgt=${cfg%%:*}; s=${cfg#*:}
  v=$(RSU_WGT_GEN=$wgt RSU_SPLIT_CHIP=$s python bench.py --steps 40 --warmup 5 --no_cpu_baseline --sustain_seconds 0 2>/dev/null | grep -o '"value": [0-9.]*' | head -1)
  echo "RSU_WGT_GEN=$wgt RSU_SPLIT_CHIP=$s $v"
done
done | tee gpurun_out/r03/split_sweep_wgt.txt
