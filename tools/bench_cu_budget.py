#!/usr/bin/env python3
"""bench.py with the conv launches planned for fewer CUs (developer tool: how much of the chip does the step need?).
usage: python tools/bench_cu_budget.py <ncu> [bench.py args]"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from road_segmentation_unet_amd._lib import call  # noqa: E402

call("rsu_set_cu_budget", int(sys.argv[1]))
sys.argv = [os.path.join(ROOT, "bench.py")] + (sys.argv[2:] or ["--steps", "10", "--warmup", "3", "--no_cpu_baseline"])
runpy.run_path(sys.argv[0], run_name="__main__")
