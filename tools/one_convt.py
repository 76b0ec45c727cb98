#!/usr/bin/env python3
"""Run one transposed-conv layer a few times (for rocprofv3 PMC passes). usage: one_convt.py H Cin Cout [op=fwd|bwd|wg] [B=4]"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from road_segmentation_unet_amd._lib import call, lib
h, cin, cout = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
op = sys.argv[4] if len(sys.argv) > 4 else "fwd"
B = int(sys.argv[5]) if len(sys.argv) > 5 else 4
DEV = "cuda:0"
ptr = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
x = torch.randn((B, h, h, cin), device=DEV).to(torch.bfloat16)
dy = torch.randn((B, 2 * h, 2 * h, cout), device=DEV).to(torch.bfloat16)
y, dx = torch.zeros_like(dy), torch.zeros_like(x)
K = torch.randn((2, 2, cout, cin), device=DEV) * 0.05
b = torch.zeros(cout, device=DEV)
pf = torch.zeros(4 * lib().rsu_packed_bytes(1, cout, (ctypes.c_int * 1)(cin), 1) // 2, dtype=torch.bfloat16, device=DEV)
pb = torch.zeros(lib().rsu_packed_bytes(4, cin, (ctypes.c_int * 1)(cout), 1) // 2, dtype=torch.bfloat16, device=DEV)
call("rsu_pack_convT_fwd", ptr(K), ptr(pf), cin, cout, st)
call("rsu_pack_convT_bwd", ptr(K), ptr(pb), cin, cout, st)
dK = torch.zeros_like(K)
ws = torch.zeros(lib().rsu_convT2x2_bwd_weight_ws_floats(cin, cout), device=DEV)
for _ in range(5):
    if op == "fwd":
        call("rsu_convT2x2_fwd", ptr(x), ptr(pf), ptr(b), ptr(y), B, h, h, cin, cout, 0, st)
    elif op == "bwd":
        call("rsu_convT2x2_bwd_data", ptr(dy), ptr(pb), ptr(dx), ptr(x), 1.0, B, h, h, cin, cout, 0, st)
    else:
        call("rsu_convT2x2_bwd_weight", ptr(x), ptr(dy), ptr(dK), None, ptr(ws), B, h, h, cin, cout, 0, st)
torch.cuda.synchronize()
