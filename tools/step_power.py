#!/usr/bin/env python3
"""Developer tool (GPU box): socket power (the device's hwmon power1_input, sampled every 50 ms) while the c2 training step runs back to back for a few
seconds -- the whole step, the forward pass alone, and forward + backward without the update -- next to the step's rate: the figures behind DESIGN.md 7.1.
usage: python tools/step_power.py [seconds=5] [workload=c2|c3|c4]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from road_segmentation_unet_amd.unet import UNet  # noqa: E402
from tools.corun_split import Power  # noqa: E402

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 5.0
wl = sys.argv[2] if len(sys.argv) > 2 else "c2"
L, dil, B = {"c2": (5, False, 4), "c3": (6, True, 1), "c4": (6, False, 4)}[wl]
m = UNet(L, 64, dil, B, 388, training=True, seed=2018)
g = torch.Generator(device="cpu").manual_seed(2017)
m.x.copy_(torch.rand((B, m.S, m.S, 3), generator=g))
m.labels.copy_((torch.rand((B, 388, 388), generator=g) < 0.2).to(torch.int64))
m.tune()
inv = 1.0 / (B * 388 * 388)


def step():
    m.forward_device(); m.backward_device(inv); m.apply_momentum(0.0, 0.9)


def fwd():
    m.forward_device()


def fwdbwd():
    m.forward_device(); m.backward_device(inv)


power = Power()
power.start()
print("power source:", power.path or "rocm-smi")
time.sleep(1.0)
power.begin(); time.sleep(2.0); idle, n0 = power.end()
print("idle (context up, nothing running): %.0f W" % idle)
for name, fn in (("whole step", step), ("forward only", fwd), ("forward + backward, no update", fwdbwd)):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    power.begin()
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < secs:
        for _ in range(20):
            fn()
        torch.cuda.synchronize()   # (every 20 passes: the host stays a few passes ahead, the queue never runs dry for long)
        n += 20
    dt = time.perf_counter() - t0
    w, ns = power.end()
    print("%-32s %7.3f ms per pass  %7.1f patches/s  %6.0f W (median of %d samples)  %.2f J per pass" % (name, dt / n * 1e3, B * n / dt, w, ns, w * dt / n))
power.stop_ = True
