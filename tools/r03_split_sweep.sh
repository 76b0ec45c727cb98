mkdir -p gpurun_out/r03
for rep in 1 2; do
for s in "128,128" "136,120" "120,136" "144,112" "112,144"; do
  v=$(RSU_SPLIT_CHIP=$s python bench.py --steps 30 --warmup 5 --no_cpu_baseline --sustain_seconds 0 2>/dev/null | grep -o '"value": [0-9.]*' | head -1)
  echo "RSU_SPLIT_CHIP=$s $v"
done
done | tee gpurun_out/r03/split_sweep.txt
