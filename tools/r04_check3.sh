#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/r04; mkdir -p $OUT; cd $REPO
timeout 1500 python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_cfg_matrix.py -x -q -m gpu -k "split_k or conv2d" 2>&1 | tail -15 > $OUT/check3_tests.txt
cat $OUT/check3_tests.txt
timeout 900 python3 __graft_entry__.py smoke 2>&1 | tail -2
timeout 600 python3 bench.py --workload c3 --no_cpu_baseline > $OUT/bench_c3_split.json 2> $OUT/bench_c3_split.err; tail -2 $OUT/bench_c3_split.err
RSU_KSPLIT=0 timeout 600 python3 bench.py --workload c3 --no_cpu_baseline > $OUT/bench_c3_nosplit.json 2> $OUT/bench_c3_nosplit.err
RSU_COB_GROUP=0 timeout 600 python3 bench.py --workload c3 --no_cpu_baseline > $OUT/bench_c3_split_cg0.json 2> $OUT/bench_c3_split_cg0.err
python3 - <<'PY'
import json,os
for f in ("bench_c3_split.json","bench_c3_nosplit.json","bench_c3_split_cg0.json"):
    try:
        d=json.loads(open(os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out/r04",f)).read().strip().splitlines()[-1])
        r=d["roofline"]
        print(f, round(d["value"],1), round(r["frac"],4), {k:(round(v["tflops"]),round(v["wall_ms_per_step"],3)) for k,v in r["by_kernel"].items()})
    except Exception as e: print(f, "failed", e)
PY
RSU_PLAN_DEBUG=1 timeout 600 python3 bench.py --workload c3 --steps 2 --warmup 0 --no_cpu_baseline --sustain_seconds 0 2>&1 | grep "plan fwd2" | sort | uniq -c | sort -k5,5 -k6,6n | cut -c1-230 > $OUT/tile_util_c3.txt
grep -v "ksplit1 " $OUT/tile_util_c3.txt | head -40
