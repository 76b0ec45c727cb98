#!/bin/bash
# Runs on the GPU box (via gpurun): bench + rocprofv3 kernel-trace stats of the same command.
# usage: tools/gpu_bench_profile.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
python3 bench.py --steps 10 --warmup 3 "$@" > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err
echo "bench rc=$?"; tail -c 3000 $OUT/bench_$TAG.json; tail -5 $OUT/bench_$TAG.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o prof -- python3 $REPO/bench.py --steps 5 --warmup 2 --no_cpu_baseline "$@" > $OUT/prof_$TAG.log 2>&1
echo "rocprof rc=$?"
find $OUT/prof_$TAG -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -40 {}'
# keep only the small summaries (the full trace can be large)
find $OUT/prof_$TAG -name "*kernel_trace.csv" -size +20M -delete
