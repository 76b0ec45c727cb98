"""CPU: `python bench.py --gpus N` without WORLD_SIZE starts its own ranks (bench.launch_ranks: torch.distributed.run as a child
process, 127.0.0.1 rendezvous, output and exit code relayed). Exercised here with a stand-in rank program over gloo, world size 2;
the real thing -- bench.py itself, two ranks on one GPU -- is tests/test_gpu_bench_contract.py::test_bench_bare_gpus_2_starts_its_own_ranks."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HELPER = os.path.join(ROOT, "tests", "helper_rank_echo.py")
SNIPPET = "import sys; sys.path.insert(0, %r); import bench; sys.exit(bench.launch_ranks(2, %r, %r, timeout=300))"


def _run(extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    return subprocess.run([sys.executable, "-c", SNIPPET % (ROOT, HELPER, ["--steps", "3"] + extra)], capture_output=True, text=True,
                          timeout=600, cwd=ROOT, env=env)


def test_launcher_starts_two_ranks_and_relays_their_line():
    out = _run([])
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["world_size"] == 2 and d["sum"] == 3.0 and d["args"] == ["--steps", "3"] and d["ipc_legacy"] == "0"


def test_launcher_relays_a_failing_rank():
    out = _run(["--fail"])
    assert out.returncode != 0


def test_bench_main_takes_the_launcher_path_before_touching_the_gpu():
    """bare --gpus 2 in this GPU-less container: the ranks are started (and die on the missing HIP device, loudly) -- the old code
    asserted `world == args.gpus` in the first lines instead"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["RSU_BENCH_BACKEND"] = "gloo"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no_cpu_baseline"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    import torch
    if torch.cuda.is_available():
        assert out.returncode == 0, out.stderr[-2000:]
        return
    assert out.returncode != 0
    assert "launch with torch.distributed.run" not in out.stderr
    assert "torch.distributed" in out.stderr or "elastic" in out.stderr or "RsuError" in out.stderr or "HIP" in out.stderr, out.stderr[-1500:]
