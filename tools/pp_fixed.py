#!/usr/bin/env python3
"""Developer tool (GPU box): forward / backward-data of single 3x3 layers at a FIXED tile shape (RSU_FWD2_CFG, no tuning, no split), primed
clocks, median of many launches -- for A/B of two library builds (RSU_LIB_PATH) without the tuner's pick in the comparison.
usage: pp_fixed.py H,Cin,Cout,cfg [H,Cin,Cout,cfg ...]"""
import ctypes, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from road_segmentation_unet_amd._lib import RsuSrc, call, lib
D = "cuda:0"; B = 4
ptr = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
os.environ["RSU_AUTOTUNE"] = "0"; os.environ["RSU_KSPLIT"] = "0"
out = []
for spec in sys.argv[1:]:
    H, cin, cout, cfg = [int(v) for v in spec.split(",")]
    os.environ["RSU_FWD2_CFG"] = str(cfg)
    ho = H - 2
    x = torch.randn((B, H, H, cin), device=D).to(torch.bfloat16); dz = torch.randn((B, ho, ho, cout), device=D).to(torch.bfloat16)
    y = torch.zeros((B, ho, ho, cout), device=D, dtype=torch.bfloat16); dx = torch.zeros_like(x)
    w = torch.randn((3, 3, cin, cout), device=D) * 0.05; bias = torch.zeros(cout, device=D)
    seg = (ctypes.c_int * 1)(cin); seg2 = (ctypes.c_int * 1)(cout)
    pf = torch.zeros(lib().rsu_packed_bytes(9, cout, seg, 1) // 2, dtype=torch.bfloat16, device=D)
    pb = torch.zeros(lib().rsu_packed_bytes(9, cin, seg2, 1) // 2, dtype=torch.bfloat16, device=D)
    call("rsu_pack_conv_fwd", ptr(w), ptr(pf), 3, cin, cout, seg, 1, st); call("rsu_pack_conv_bwd", ptr(w), ptr(pb), 3, cin, 0, cin, cout, st)
    arr = (RsuSrc * 1)(RsuSrc(x.data_ptr(), H, H, cin, 0, 0))
    fwd = lambda: call("rsu_conv2d_fwd", arr, 1, ptr(pf), ptr(bias), ptr(y), B, H, H, cout, 1, 1, 0, st)
    bwd = lambda: call("rsu_conv2d_bwd_data", ptr(dz), ptr(pb), ptr(dx), ptr(x), 0, B, H, H, cin, 0, cin, cout, 1, 0, st)
    res = []
    for f in (fwd, bwd):
        t0 = time.time()
        while time.time() - t0 < 0.3: f()
        torch.cuda.synchronize()
        ts = []
        for _ in range(60):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); f(); b.record(); b.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
        res.append(float(np.median(ts)))
    out.append("%s fwd %.1f bwd %.1f" % (spec, res[0], res[1]))
print(" | ".join(out))
