#!/bin/bash
# GPU box: the whole -m gpu suite (parity figures -> gpurun_out/r05/parity_record.json), then the c3 step's timeline and bench lines
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/r05; mkdir -p $OUT; cd $REPO
rm -f $OUT/parity_record.json; export RSU_PARITY_RECORD=$OUT/parity_record.json
timeout 2400 python3 -m pytest tests -q -x -m gpu 2>&1 | grep -v amdgpu.ids | tail -8 > $OUT/gputests.txt; cat $OUT/gputests.txt
unset RSU_PARITY_RECORD
python3 bench.py --workload c3 --no_cpu_baseline > $OUT/bench_c3_start.json 2>/dev/null; cut -c1-200 $OUT/bench_c3_start.json
EXTRA_BENCH="--workload c3" bash tools/timeline.sh c3 > /dev/null 2>&1
head -5 $OUT/timeline_c3.txt
