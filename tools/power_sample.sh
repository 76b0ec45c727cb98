#!/bin/bash
# usage (GPU box): tools/power_sample.sh -> gpurun_out/r03/power_during_bench.txt: rocm-smi power / clocks sampled while bench.py's sustained loop runs
mkdir -p gpurun_out/r03
python bench.py --steps 20 --warmup 3 --no_cpu_baseline --sustain_seconds 12 > /tmp/pb.json 2>/dev/null &
BP=$!
sleep 6
for i in 1 2 3 4 5 6; do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk|fclk" | tr -s ' ' | sed 's/^/  /'
  echo "  --"
  sleep 1.5
done > gpurun_out/r03/power_during_bench.txt
wait $BP
grep -o '"sustained": {[^}]*}' /tmp/pb.json >> gpurun_out/r03/power_during_bench.txt
cat gpurun_out/r03/power_during_bench.txt
