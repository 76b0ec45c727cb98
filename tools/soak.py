#!/usr/bin/env python3
"""Soak / race check (developer tool, GPU box): N training steps of config 2 from the same state twice; the kernels' reduction
orders are fixed, so the two weight vectors must be bit-identical. usage: python tools/soak.py [steps=100] [keep=1.0]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from road_segmentation_unet_amd.unet import UNet
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
keep = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
def run():
    m = UNet(5, 64, False, 4, 388, seed=2018, training=True)
    g = torch.Generator(device="cpu").manual_seed(7)
    losses = []
    for s in range(steps):
        m.x.copy_(torch.rand((4, m.S, m.S, 3), generator=g))
        m.labels.copy_((torch.rand((4, 388, 388), generator=g) < 0.2).to(torch.int64))
        m.forward_device(keep=keep)
        m.backward_device(1.0 / (4 * 388 * 388))
        m.apply_momentum(0.01, 0.9)
        if s % 25 == 0 or s == steps - 1:
            losses.append(float(m.loss_sum.item()) / (4 * 388 * 388))
    torch.cuda.synchronize()
    return m.flat_w.clone(), losses
t0 = time.time()
w1, l1 = run()
w2, l2 = run()
print("steps", steps, "keep", keep, "time %.1fs" % (time.time() - t0))
print("losses", ["%.5f" % v for v in l1])
print("finite", bool(torch.isfinite(w1).all()), "bit-identical runs", bool(torch.equal(w1, w2)), "max |dw|", float((w1 - w2).abs().max()))
assert torch.isfinite(w1).all() and torch.equal(w1, w2) and l1 == l2
