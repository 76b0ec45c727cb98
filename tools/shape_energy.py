#!/usr/bin/env python3
"""Developer tool (GPU box): time, socket power and ENERGY per launch of one 3x3 layer (forward) under every admissible tile shape (RSU_FWD2_CFG forced,
no tuning, no split-K), each held for ~2 s -- under a power envelope the cheaper shape, not the faster-alone one, is the one to pick.
usage: python tools/shape_energy.py [H Cin Cout [B]] ..."""
import ctypes, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from road_segmentation_unet_amd._lib import RsuSrc, call, lib  # noqa: E402
from tools.corun_split import Power  # noqa: E402
D = "cuda:0"
ptr = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
NAMES = ["128x256", "64x512", "128x128", "64x256", "128x192", "64x384", "128x320", "64x640"]
layers = [(282, 128, 128, 4), (570, 64, 64, 4), (138, 256, 256, 4), (66, 512, 512, 4)]
if len(sys.argv) >= 4:
    layers = [(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]) if len(sys.argv) > 4 else 4)]
os.environ["RSU_AUTOTUNE"] = "0"
power = Power(); power.start()
for H, cin, cout, B in layers:
    ho = H - 2
    x = torch.randn((B, H, H, cin), device=D).relu().to(torch.bfloat16)
    y = torch.zeros((B, ho, ho, cout), device=D, dtype=torch.bfloat16)
    w = torch.randn((3, 3, cin, cout), device=D) * 0.05
    bias = torch.zeros(cout, device=D)
    seg = (ctypes.c_int * 1)(cin)
    pf = torch.zeros(lib().rsu_packed_bytes(9, cout, seg, 1) // 2, dtype=torch.bfloat16, device=D)
    call("rsu_pack_conv_fwd", ptr(w), ptr(pf), 3, cin, cout, seg, 1, st)
    arr = (RsuSrc * 1)(RsuSrc(x.data_ptr(), H, H, cin, 0, 0))
    fl = 2.0 * B * ho * ho * cout * cin * 9
    print("== forward %d px, %d -> %d channels, batch %d (%.1f GFLOP per launch)" % (H, cin, cout, B, fl / 1e9), flush=True)
    for cfg in range(8):
        os.environ["RSU_FWD2_CFG"] = str(cfg)
        try:
            for _ in range(3):
                call("rsu_conv2d_fwd", arr, 1, ptr(pf), ptr(bias), ptr(y), B, H, H, cout, 1, 1, 0, st)
            torch.cuda.synchronize()
        except Exception:
            print("  %-8s n/a" % NAMES[cfg]); continue
        power.begin(); t0 = time.perf_counter(); n = 0
        while time.perf_counter() - t0 < 2.0:
            for _ in range(200):
                call("rsu_conv2d_fwd", arr, 1, ptr(pf), ptr(bias), ptr(y), B, H, H, cout, 1, 1, 0, st)
            torch.cuda.synchronize(); n += 200
        dt = time.perf_counter() - t0
        wv, ns = power.end()
        print("  %-8s %7.1f us  %6.0f TFLOP/s  %5.0f W  %6.1f mJ per launch  (%.2f pJ per FLOP)" % (NAMES[cfg], dt / n * 1e6, fl / (dt / n) / 1e12, wv, wv * dt / n * 1e3, wv * dt / n / fl * 1e12), flush=True)
    os.environ.pop("RSU_FWD2_CFG", None)
power.stop_ = True
