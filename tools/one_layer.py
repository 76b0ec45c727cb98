#!/usr/bin/env python3
"""Run one conv op repeatedly (for rocprofv3 --pmc). usage: one_layer.py H Cin Cout [op=fwd|bwd|wg] [B=4] [reps=20] [dil=1]"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from road_segmentation_unet_amd._lib import RsuSrc, call, lib
H, cin, cout = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
op = sys.argv[4] if len(sys.argv) > 4 else "fwd"
B = int(sys.argv[5]) if len(sys.argv) > 5 else 4
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 20
dil = int(sys.argv[7]) if len(sys.argv) > 7 else 1
D = "cuda:0"
ptr = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
ho = H - 2 * dil
x = torch.randn((B, H, H, cin), device=D).to(torch.bfloat16)
dz = torch.randn((B, ho, ho, cout), device=D).to(torch.bfloat16)
y = torch.zeros((B, ho, ho, cout), device=D, dtype=torch.bfloat16)
dx = torch.zeros_like(x)
w = torch.randn((3, 3, cin, cout), device=D) * 0.05
bias = torch.zeros(cout, device=D)
seg = (ctypes.c_int * 1)(cin)
pf = torch.zeros(lib().rsu_packed_bytes(9, cout, seg, 1) // 2, dtype=torch.bfloat16, device=D)
seg2 = (ctypes.c_int * 1)(cout)
pb = torch.zeros(lib().rsu_packed_bytes(9, cin, seg2, 1) // 2, dtype=torch.bfloat16, device=D)
call("rsu_pack_conv_fwd", ptr(w), ptr(pf), 3, cin, cout, seg, 1, st)
call("rsu_pack_conv_bwd", ptr(w), ptr(pb), 3, cin, 0, cin, cout, st)
dw = torch.zeros_like(w)
ws = torch.zeros(lib().rsu_conv2d_bwd_weight_ws_floats(cin, cin, cout), device=D)
src = RsuSrc(x.data_ptr(), H, H, cin, 0, 0)
arr = (RsuSrc * 1)(src)
for _ in range(reps):
    if op == "fwd":
        call("rsu_conv2d_fwd", arr, 1, ptr(pf), ptr(bias), ptr(y), B, H, H, cout, dil, 1, 0, st)
    elif op == "bwd":
        call("rsu_conv2d_bwd_data", ptr(dz), ptr(pb), ptr(dx), ptr(x), 0, B, H, H, cin, 0, cin, cout, dil, 0, st)
    else:
        call("rsu_conv2d_bwd_weight", ctypes.byref(src), ptr(dz), ptr(dw), None, ptr(ws), B, ho, ho, cin, 0, cout, dil, 0, st)
torch.cuda.synchronize()
