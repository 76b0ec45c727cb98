"""-m gpu: bench.py's one-line JSON contract (the driver parses it): metric / value / unit / steps / roofline / cpu_baseline."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_bench_prints_one_contract_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--cpu_sample_patch", "68"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["metric"] == base["metric"] and d["unit"] == "patches/s"
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["dtype"] == "bf16" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert d["value"] > 50 and abs(d["value"] - 4 * 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
    assert "num_layers=5 root_size=64 patch_size=388" in d["config"]["workload"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["measured_in"].startswith("the schedule `value` is timed on") and 0 < r["frac_serial_per_layer"] < 1 and 0 < r["frac_single_stream_grouped"] < 1
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 2500.0
    assert 0 < r["achieved"] < r["peak"] and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["traffic"] is None or r["traffic"] > 1e6
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["unit"] == "patches/s" and c["value"] > 0 and c["cores"] >= 1 and c["sample"]


def test_bench_two_ranks_over_gloo_on_one_gpu():
    """The driver launches bench.py under torch.distributed.run for N > 1; here the same launch with two ranks on the one GPU
    of the test box (gloo instead of RCCL): sharded step, schedule tuning, MAX-over-ranks timing, one line from rank 0."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, RSU_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                          "--no_cpu_baseline"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, "\n".join(ln for ln in out.stderr.splitlines() if "amdgpu.ids" not in ln and "site-packages" not in ln)[-8000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 8 and d["config"]["parallelism"] == "dp2"
    assert d["value"] > 0.5 and abs(d["value"] - 8 * 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
    ex = d["config"]["dp_exchange"]
    assert ex["backward_cu_budget"] in (256, 240, 224, 208) and len(ex["tuned_ms_per_step"]) == 6 and ex["min_bucket_MB"] in (4, 32)
    assert "cpu_baseline" not in d or d["cpu_baseline"] is None or d["cpu_baseline"]


def test_bench_bare_gpus_2_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no WORLD_SIZE (how the driver starts the N = 1 run): bench.py launches its two ranks itself as a
    child torch.distributed.run (gloo here: the test box has one GPU) and relays rank 0's contract line"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(RSU_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    # (round 4: two-rank runs failed once in ~4 with "connection reset by peer" -- the repetition of an implausible instrumented pass was
    # decided per rank, so one rank ran a pass, collectives included, more than its peer; the decision is an all-reduce now)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no_cpu_baseline",
                          "--sustain_seconds", "0"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, "\n".join(ln for ln in out.stderr.splitlines() if "amdgpu.ids" not in ln and "site-packages" not in ln)[-8000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["dp_exchange"]["exchange_proof"]["world_size"] == 2
    assert d["roofline"]["measured_in"].startswith("the schedule `value` is timed on")


def test_bench_workload_c3_line():
    """--workload c3: BASELINE configs[2] (num_layers=6, dilated, one patch per step) through the same contract"""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "c3", "--steps", "3", "--warmup", "1", "--no_cpu_baseline",
                          "--sustain_seconds", "0"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")][0])
    assert "num_layers=6" in d["config"]["workload"] and "dilated" in d["config"]["workload"] and d["config"]["batch_per_gpu"] == 1
    assert d["value"] > 20 and 0 < d["roofline"]["frac"] < 1
