#!/usr/bin/env python3
"""conv2 + max-pool of the encoder levels of config 2: rsu_conv2d_fwd_pool with the pool folded into the conv epilogue against the two
launches (developer tool, GPU box). usage: python tools/bench_pool.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from road_segmentation_unet_amd._lib import RsuSrc, call, lib  # noqa: E402

DEV = "cuda:0"


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def timeit(fn, reps=8):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
call("rsu_set_autotune", 2)
B = 4
for H, C in ((570, 64), (282, 128), (138, 256), (66, 512)):
    x = torch.randn((B, H, H, C), device=DEV).to(torch.bfloat16)
    w = torch.randn((3, 3, C, C), device=DEV) * 0.05
    bias = torch.zeros(C, device=DEV)
    seg = (ctypes.c_int * 1)(C)
    pf = torch.zeros(lib().rsu_packed_bytes(9, C, seg, 1) // 2, dtype=torch.bfloat16, device=DEV)
    call("rsu_pack_conv_fwd", ptr(w), ptr(pf), 3, C, C, seg, 1, st)
    Ho = H - 2
    y = torch.zeros((B, Ho, Ho, C), device=DEV, dtype=torch.bfloat16)
    pool = torch.zeros((B, Ho // 2, Ho // 2, C), device=DEV, dtype=torch.bfloat16)
    code = torch.zeros((B, Ho // 2, Ho // 2, C), device=DEV, dtype=torch.uint8)
    arr = (RsuSrc * 1)(RsuSrc(x.data_ptr(), H, H, C, 0, 0))
    res = {}
    for fused in ("1", "0"):
        os.environ["RSU_POOL_FUSED"] = fused
        res[fused] = timeit(lambda: call("rsu_conv2d_fwd_pool", arr, 1, ptr(pf), ptr(bias), ptr(y), ptr(pool), ptr(code), B, H, H, C, 1.0, 0, 0, st))
    tconv = timeit(lambda: call("rsu_conv2d_fwd", arr, 1, ptr(pf), ptr(bias), ptr(y), B, H, H, C, 1, 1, 0, st))
    os.environ["RSU_PLAN_DEBUG"] = "1"
    os.environ["RSU_POOL_FUSED"] = "1"
    call("rsu_conv2d_fwd_pool", arr, 1, ptr(pf), ptr(bias), ptr(y), ptr(pool), ptr(code), B, H, H, C, 1.0, 0, 0, st)
    os.environ.pop("RSU_PLAN_DEBUG")
    print("H %4d C %4d: fused %.1f us | conv + pool kernel %.1f us | conv alone %.1f us" % (H, C, res["1"], res["0"], tconv), flush=True)
