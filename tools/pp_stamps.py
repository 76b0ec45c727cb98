#!/usr/bin/env python3
"""Developer tool (GPU box): run one 3x3 conv layer on the time-stamping build of igemm_pp and print how long the barrier
intervals of a workgroup last. usage: pp_stamps.py H Cin Cout [op=fwd|bwd] [B=4] [cfg=0|1] [dbg bits to add]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from road_segmentation_unet_amd._lib import RsuSrc, call, lib  # noqa: E402

H, cin, cout = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
op = sys.argv[4] if len(sys.argv) > 4 else "fwd"
B = int(sys.argv[5]) if len(sys.argv) > 5 else 4
cfg = int(sys.argv[6]) if len(sys.argv) > 6 else 0
extra = int(sys.argv[7]) if len(sys.argv) > 7 else 0
zero = len(sys.argv) > 8 and sys.argv[8] == "zero"
NST = 640
D = "cuda:0"
ptr = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
ho = H - 2
x = torch.randn((B, H, H, cin), device=D).to(torch.bfloat16)
dz = torch.randn((B, ho, ho, cout), device=D).to(torch.bfloat16)
y = torch.zeros((B, ho, ho, cout), device=D, dtype=torch.bfloat16)
dx = torch.zeros_like(x)
w = torch.randn((3, 3, cin, cout), device=D) * 0.05
if zero:
    x.zero_(); dz.zero_(); w.zero_()
bias = torch.zeros(cout, device=D)
seg = (ctypes.c_int * 1)(cin)
pf = torch.zeros(lib().rsu_packed_bytes(9, cout, seg, 1) // 2, dtype=torch.bfloat16, device=D)
seg2 = (ctypes.c_int * 1)(cout)
pb = torch.zeros(lib().rsu_packed_bytes(9, cin, seg2, 1) // 2, dtype=torch.bfloat16, device=D)
call("rsu_pack_conv_fwd", ptr(w), ptr(pf), 3, cin, cout, seg, 1, st)
call("rsu_pack_conv_bwd", ptr(w), ptr(pb), 3, cin, 0, cin, cout, st)
arr = (RsuSrc * 1)(RsuSrc(x.data_ptr(), H, H, cin, 0, 0))
stamps = torch.zeros((256 * 8 * NST,), dtype=torch.int32, device=D)
os.environ["RSU_AUTOTUNE"] = "0"
os.environ["RSU_FWD2_CFG"] = str(cfg)


def run():
    if op == "fwd":
        call("rsu_conv2d_fwd", arr, 1, ptr(pf), ptr(bias), ptr(y), B, H, H, cout, 1, 1, 0, st)
    else:
        call("rsu_conv2d_bwd_data", ptr(dz), ptr(pb), ptr(dx), ptr(x), 0, B, H, H, cin, 0, cin, cout, 1, 0, st)


for _ in range(3):
    run()
os.environ["RSU_FWD_DBG"] = str(128 | extra)
os.environ["RSU_STAMP_PTR"] = hex(stamps.data_ptr())
for _ in range(3):
    run()
torch.cuda.synchronize()
s = stamps.cpu().numpy().astype(np.int64).reshape(256, 8, NST) & 0xFFFFFFFF
clk = s[:, 0, NST - 2].astype(np.float64)
rt = s[:, 0, NST - 1].astype(np.float64)
ok = rt > 0
print("in-kernel clock (shader cycles / 100-MHz ticks, per workgroup): median %.3f GHz, min %.3f, max %.3f; loop length median %.1f us"
      % (np.median(clk[ok] / rt[ok]) * 0.1, (clk[ok] / rt[ok]).min() * 0.1, (clk[ok] / rt[ok]).max() * 0.1, np.median(rt[ok]) * 0.01))
seg = s[:, :, NST - 10:NST - 2].astype(np.float64)
names = ["epilogue (convert + store)", "tile decode", "bias init", "fragment reads (issue)", "prefetch issues", "waits (vmcnt, lgkmcnt)", "epilogue (addresses)"]
for wv in (0, 4):
    print("wave %d, cycles per workgroup in R-interval segments (median over workgroups): " % wv +
          ", ".join("%s %.0f" % (names[k], np.median(seg[ok, wv, k])) for k in range(7)) + "; loop %.0f" % np.median(clk[ok]))
s[:, :, NST - 10:] = 0
# stamps per phase of a wave: [end of R work] barrier [start of M] ... [end of M issue] barrier [start of next R]
for blk in (0, 100):
    t = s[blk]
    n = int((t[0] != 0).sum()) // 4 * 4
    if n < 160:
        continue
    for wv in (0, 1, 4):
        q = t[wv, :n].reshape(-1, 4)   # columns: Rend, Mstart, Mend, Rstart(next)
        r_work = (q[1:, 0] - q[:-1, 3]) & 0xFFFFFFFF     # start of R .. reads landed + bookkeeping done
        r_wait = (q[:, 1] - q[:, 0]) & 0xFFFFFFFF        # waiting at the barrier that ends R
        m_work = (q[:, 2] - q[:, 1]) & 0xFFFFFFFF        # MFMA issue
        m_wait = (q[:, 3] - q[:, 2]) & 0xFFFFFFFF        # waiting at the barrier that ends M
        print("block %d wave %d: %d phases | R work mean %.0f med %.0f p90 %.0f | R barrier wait mean %.0f med %.0f | M work mean %.0f med %.0f p90 %.0f | M barrier wait mean %.0f med %.0f"
              % (blk, wv, len(q), r_work.mean(), np.median(r_work), np.percentile(r_work, 90), r_wait.mean(), np.median(r_wait),
                 m_work.mean(), np.median(m_work), np.percentile(m_work, 90), m_wait.mean(), np.median(m_wait)))
        if wv in (0, 4):
            k0 = 18
            print("   phases %d..%d  Rwork:" % (k0, k0 + 18), " ".join("%d" % v for v in r_work[k0:k0 + 18]))
            print("                  Rwait:", " ".join("%d" % v for v in r_wait[k0 + 1:k0 + 19]))
            print("                  Mwork:", " ".join("%d" % v for v in m_work[k0 + 1:k0 + 19]))
            print("                  Mwait:", " ".join("%d" % v for v in m_wait[k0 + 1:k0 + 19]))
