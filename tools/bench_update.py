#!/usr/bin/env python3
"""Developer tool (GPU box): the Momentum + re-pack pass (rsu_update_table_run) alone: us per call and GB/s at 24 B per parameter, for the c2 and the c3
network. usage: bench_update.py [c2] [c3]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from road_segmentation_unet_amd.unet import UNet
for wl in (sys.argv[1:] or ["c2", "c3"]):
    L, dil, B = (5, False, 4) if wl == "c2" else (6, True, 1)
    m = UNet(L, 64, dil, B, 388, training=True)
    m.flat_g.normal_(0, 1e-3)
    for _ in range(5): m.apply_momentum(0.0, 0.9)
    torch.cuda.synchronize()
    ts = []
    for _ in range(30):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); m.apply_momentum(0.0, 0.9); b.record(); b.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    us = ts[len(ts) // 2]
    print("%s: %d live parameters, update %.1f us (median of 30) = %.2f TB/s at 24 B per parameter" % (wl, m.n_live, us, m.n_live * 24 / us / 1e6))
    del m
