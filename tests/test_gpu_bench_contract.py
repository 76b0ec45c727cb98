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
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 2500.0
    assert 0 < r["achieved"] < r["peak"] and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["traffic"] is None or r["traffic"] > 1e6
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["unit"] == "patches/s" and c["value"] > 0 and c["cores"] >= 1 and c["sample"]
