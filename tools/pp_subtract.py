#!/usr/bin/env python3
"""Developer tool (GPU box, DEV library): igemm_pp with parts of its work switched off (RSU_FWD_DBG timing bits, results are wrong by design):
what each part costs the layer once everything else is in place. usage: pp_subtract.py H Cin Cout [op] [B] [cfg]"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from road_segmentation_unet_amd._lib import RsuSrc, call, lib
H, cin, cout = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
op = sys.argv[4] if len(sys.argv) > 4 else "fwd"
B = int(sys.argv[5]) if len(sys.argv) > 5 else 4
cfg = int(sys.argv[6]) if len(sys.argv) > 6 else 0
D = "cuda:0"
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
os.environ["RSU_AUTOTUNE"] = "0"; os.environ["RSU_FWD2_CFG"] = str(cfg); os.environ["RSU_KSPLIT"] = "0"
def run():
    if op == "fwd": call("rsu_conv2d_fwd", arr, 1, ptr(pf), ptr(bias), ptr(y), B, H, H, cout, 1, 1, 0, st)
    else: call("rsu_conv2d_bwd_data", ptr(dz), ptr(pb), ptr(dx), ptr(x), 0, B, H, H, cin, 0, cin, cout, 1, 0, st)
def timed(dbg, n=30):
    os.environ["RSU_FWD_DBG"] = str(dbg)
    for _ in range(5): run()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); run(); b.record(); b.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
    return float(np.median(ts))
gf = 2.0 * B * ho * ho * cin * cout * 9 / 1e9
CASES = [(0, "production kernel"), (256, "DBG build, prefetch stream never changes tile"), (64 | 256, "no tile change at all"), (16, "no bias start"),
         (512, "epilogue without its stores"), (8, "no epilogue"), (8 | 16 | 64 | 256, "no tile boundary work (epilogue, bias, tile change)"),
         (1, "no weight DMA"), (2, "no halo DMA"), (3, "no DMA"), (3 | 8 | 16 | 64 | 256, "fragment reads + MFMAs + barriers only"),
         (32, "DMA streams + boundaries, no fragment reads / MFMAs"), (32 | 8 | 16 | 64 | 256, "DMA streams alone")]
base = None
print("igemm_pp %s H=%d %d->%d B=%d cfg=%d: %.1f GFLOP (MFMA floor at 2.5 PFLOP/s: %.1f us)" % (op, H, cin, cout, B, cfg, gf, gf / 2.5e3))
for dbg, what in CASES:
    t = timed(dbg)
    if base is None: base = t
    print("  dbg %4d  %7.1f us  %+6.1f us   %s" % (dbg, t, t - base, what))
