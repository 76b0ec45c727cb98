#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/r04; mkdir -p $OUT; cd $REPO
timeout 1500 python3 -m pytest tests/test_gpu_bench_contract.py tests/test_bench_launcher.py "tests/test_gpu_net.py::test_c2_benchmarked_batch_of_four_equals_one_patch" -x -q -m "gpu or not gpu" 2>&1 | tail -15 > $OUT/check1_tests.txt
timeout 600 python3 bench.py --no_cpu_baseline > $OUT/bench_check1.json 2> $OUT/bench_check1.err
timeout 600 python3 bench.py --workload c3 --no_cpu_baseline > $OUT/bench_c3_before.json 2> $OUT/bench_c3_before.err
cat $OUT/check1_tests.txt; python3 - <<'PY'
import json,os
for f in ("bench_check1.json","bench_c3_before.json"):
    try:
        d=json.loads(open(os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out/r04",f)).read().strip().splitlines()[-1])
        r=d["roofline"]
        print(f, d["value"], r["frac"], r["whole_step_frac"], r["frac_serial_per_layer"], r["frac_single_stream_grouped"], {k:(round(v["tflops"]),round(v["chip_ms_per_step"],3),round(v["wall_ms_per_step"],3)) for k,v in r["by_kernel"].items()})
    except Exception as e: print(f, "failed", e)
PY
tail -3 $OUT/bench_check1.err $OUT/bench_c3_before.err
