#!/usr/bin/env python3
"""Developer tool (GPU box): the transposed conv's weight gradient per level at a CU budget, generic launch (RSU_WGT_GEN=1) against igemm_wgt (3).
usage: python tools/bench_convT_wg.py [--ncu 256,128] [--L 5 --root 64 --P 388 --B 4]"""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from road_segmentation_unet_amd._lib import call, lib  # noqa: E402
from road_segmentation_unet_amd.unet import input_size_needed  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--L", type=int, default=5)
ap.add_argument("--root", type=int, default=64)
ap.add_argument("--P", type=int, default=388)
ap.add_argument("--B", type=int, default=4)
ap.add_argument("--ncu", default="256,128")
a = ap.parse_args()
ptr = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
h = input_size_needed(a.P, a.L)
for _ in range(a.L - 1):
    h = (h - 4) // 2
h -= 4
nf = a.root * 2 ** (a.L - 1)
for i in range(a.L - 1):
    cin, cout = nf, nf // 2
    x = torch.randn((a.B, h, h, cin), device="cuda:0").to(torch.bfloat16)
    dy = torch.randn((a.B, 2 * h, 2 * h, cout), device="cuda:0").to(torch.bfloat16)
    dK = torch.zeros((2, 2, cout, cin), device="cuda:0")
    db = torch.zeros(cout, device="cuda:0")
    ws = torch.zeros(lib().rsu_convT2x2_bwd_weight_ws_floats(cin, cout), device="cuda:0")
    fl = 2.0 * a.B * h * h * cin * cout * 4
    line = "up_conv_%d  H %3d C %4d->%4d %6.1f GF |" % (i, h, cin, cout, fl / 1e9)
    for ncu in [int(v) for v in a.ncu.split(",")]:
        for gen in ("1", "3"):
            os.environ["RSU_WGT_GEN"] = gen
            fn = lambda: call("rsu_convT2x2_bwd_weight", ptr(x), ptr(dy), ptr(dK), ptr(db), ptr(ws), a.B, h, h, cin, cout, ncu, st)
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record()
            torch.cuda.synchronize()
            t = e0.elapsed_time(e1) / 20 * 1e-3
            line += " ncu%d g%s %5.0fus %4.0fTF |" % (ncu, gen, t * 1e6, fl / t / 1e12)
    print(line, flush=True)
    h, nf = 2 * h - 4, nf // 2
