#!/usr/bin/env python3
"""Developer probe (GPU box): the config-2 training step (forward + backward on two streams + update) captured once as a hipGraph
(torch.cuda.CUDAGraph) and replayed, against the eager launch sequence: ms per step, and the weights after the same number of steps."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from road_segmentation_unet_amd.unet import UNet  # noqa: E402

B, P = 4, 388


def fresh():
    m = UNet(5, 64, False, B, P, seed=2018, training=True)
    g = torch.Generator(device="cpu").manual_seed(1)
    m.x.copy_(torch.rand(tuple(m.x.shape), generator=g))
    m.labels.copy_((torch.rand(tuple(m.labels.shape), generator=g) < 0.2).to(torch.int64))
    m.tune()
    return m


def step(m):
    m.forward_device()
    m.backward_device(1.0 / (B * P * P))
    m.apply_momentum(0.01, 0.9)


def timed(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


N = 40
m1 = fresh()
for _ in range(5):
    step(m1)
ms_eager = timed(lambda: step(m1), N)
m2 = fresh()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(5):
        step(m2)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    step(m2)
torch.cuda.synchronize()
# (the capture does not execute: m2 has done 5 steps, m1 5 + N)
ms_graph = timed(g.replay, N)
print("eager %.3f ms/step (%.1f patches/s) | graph replay %.3f ms/step (%.1f patches/s)" % (ms_eager, B / ms_eager * 1e3, ms_graph, B / ms_graph * 1e3))
print("weights equal after the same number of steps:", bool(torch.equal(m1.flat_w, m2.flat_w)), "max |diff| %.3e" % float((m1.flat_w - m2.flat_w).abs().max()))
