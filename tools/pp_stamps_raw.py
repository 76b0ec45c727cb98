#!/usr/bin/env python3
"""Developer tool (GPU box, DEV library): raw barrier-interval stamps of igemm_pp for one layer: for waves 0 (G0) and 4 (G1) of two workgroups the
alternating sequence work / barrier-wait / work / ... in cycles. usage: pp_stamps_raw.py H Cin Cout [op] [B] [cfg] [count]"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from road_segmentation_unet_amd._lib import RsuSrc, call, lib
H, cin, cout = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
op = sys.argv[4] if len(sys.argv) > 4 else "fwd"
B = int(sys.argv[5]) if len(sys.argv) > 5 else 4
cfg = int(sys.argv[6]) if len(sys.argv) > 6 else 0
cnt = int(sys.argv[7]) if len(sys.argv) > 7 else 120
NST = 640; D = "cuda:0"
ptr = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
ho = H - 2
x = torch.randn((B, H, H, cin), device=D).to(torch.bfloat16); dz = torch.randn((B, ho, ho, cout), device=D).to(torch.bfloat16)
y = torch.zeros((B, ho, ho, cout), device=D, dtype=torch.bfloat16); dx = torch.zeros_like(x)
w = torch.randn((3, 3, cin, cout), device=D) * 0.05; bias = torch.zeros(cout, device=D)
seg = (ctypes.c_int * 1)(cin); seg2 = (ctypes.c_int * 1)(cout)
pf = torch.zeros(lib().rsu_packed_bytes(9, cout, seg, 1) // 2, dtype=torch.bfloat16, device=D)
pb = torch.zeros(lib().rsu_packed_bytes(9, cin, seg2, 1) // 2, dtype=torch.bfloat16, device=D)
call("rsu_pack_conv_fwd", ptr(w), ptr(pf), 3, cin, cout, seg, 1, st); call("rsu_pack_conv_bwd", ptr(w), ptr(pb), 3, cin, 0, cin, cout, st)
arr = (RsuSrc * 1)(RsuSrc(x.data_ptr(), H, H, cin, 0, 0))
stamps = torch.zeros((256 * 8 * NST,), dtype=torch.int32, device=D)
os.environ["RSU_AUTOTUNE"] = "0"; os.environ["RSU_FWD2_CFG"] = str(cfg)
def run():
    if op == "fwd": call("rsu_conv2d_fwd", arr, 1, ptr(pf), ptr(bias), ptr(y), B, H, H, cout, 1, 1, 0, st)
    else: call("rsu_conv2d_bwd_data", ptr(dz), ptr(pb), ptr(dx), ptr(x), 0, B, H, H, cin, 0, cin, cout, 1, 0, st)
for _ in range(3): run()
# (RSU_STAMP_DBG: timing-ablation bits on top of the stamps, 128x256 shape only)
os.environ["RSU_FWD_DBG"] = str(128 | int(os.environ.get("RSU_STAMP_DBG", "0"))); os.environ["RSU_STAMP_PTR"] = hex(stamps.data_ptr())
for _ in range(3): run()
torch.cuda.synchronize()
s = stamps.cpu().numpy().astype(np.int64).reshape(256, 8, NST) & 0xFFFFFFFF
clk = s[:, 0, NST - 2].astype(np.float64); rt = s[:, 0, NST - 1].astype(np.float64); ok = rt > 0
print("clock median %.3f GHz; loop median %.1f us (%.0f cycles)" % (np.median(clk[ok] / rt[ok]) * 0.1, np.median(rt[ok]) * 0.01, np.median(clk[ok])))
seg = s[:, :, NST - 10:NST - 2].astype(np.float64)
names = ["epilogue", "tile decode", "bias init", "fragment reads (issue)", "prefetch issues", "waits (vmcnt, lgkmcnt)", "epilogue addresses", "a_begin (next halo: tile / source change)"]
for wv in (0, 4):
    print("wave %d, cycles per workgroup summed over the loop (median over workgroups): " % wv + ", ".join("%s %.0f" % (names[k], np.median(seg[ok, wv, k])) for k in range(8)))
s[:, :, NST - 10:] = 0
for blk in (0, 100):
    # per barrier: interval length (release to release) and every wave's work in it (arrival - previous release); G1's barrier count is
    # aligned to G0's by the release times
    t = s[blk, :, :NST - 10]
    n = min(int((t[w] != 0).sum()) for w in range(8)) // 2 * 2
    arr = t[:, 0:n:2].astype(np.int64); rel = t[:, 1:n:2].astype(np.int64)
    nb = n // 2
    shift = [0] * 8
    for w in range(1, 8):
        shift[w] = min(range(-3, 4), key=lambda k: np.abs(rel[w, 10 + k:50 + k] - rel[0, 10:50]).sum())
    print("block %d: barrier alignment shifts %s" % (blk, shift))
    print("  bar  length | work of waves 0..7 (G0 = 0-3, G1 = 4-7)")
    for b in range(12, min(nb - 4, 12 + cnt)):
        row = []
        for w in range(8):
            bw = b + shift[w]
            row.append(int((arr[w, bw] - rel[w, bw - 1]) & 0xFFFFFFFF))
        print("  %3d  %6d | %s" % (b, int((rel[0, b] - rel[0, b - 1]) & 0xFFFFFFFF), " ".join("%5d" % v for v in row)))
