#!/usr/bin/env python3
"""debug aid: one split-K conv launch, slab decoded slice by slice against numpy partial convolutions"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RSU_PLAN_DEBUG"] = "1"
from oracle import unet_oracle as U
from tests import hiputil as hu
from road_segmentation_unet_amd._lib import RsuSrc, call, lib
N, H, W, Cin, Cout = 1, 20, 20, 512, 512
rng = np.random.RandomState(1)
x = hu.q((rng.standard_normal((N, H, W, Cin))).astype(np.float32))
w = (rng.standard_normal((3, 3, Cin, Cout)) / np.sqrt(9 * Cin)).astype(np.float32)
xd = hu.dev_bf16(x); wp = hu.pack_conv_fwd(w)
Ho, Wo = H - 2, W - 2
y = torch.full((N, Ho, Wo, Cout), float("nan"), dtype=torch.bfloat16, device=hu.DEV)
nk = int(lib().rsu_conv_splitk_ws_floats())
kws = torch.full((nk,), float("nan"), dtype=torch.float32, device=hu.DEV)
s = (RsuSrc * 1)(hu.src_of(xd, H, W))
call("rsu_conv2d_fwd_k", s, 1, hu.ptr(wp), None, hu.ptr(y), N, H, W, Cout, 1, 0, 0, hu.ptr(kws), nk, hu.stream())
torch.cuda.synchronize()
# expected plan for this shape: cfg0 (128x256: WCO 2, WPX 4, CT 4, PT 4), SW 32, TR 8, 3 tiles, ncob 4, S 4
WCO, WPX, CT, PT, SW, TR, ntile, ncob, S = 2, 4, 4, 4, 32, 8, 3, 4, 4
NST = (CT // 2) * PT
nslots = ntile * ncob
stride = nslots * 128 * 256
slab = kws[:S * stride].cpu().numpy().reshape(S, nslots, 8, NST, 2, 64, 4)
wq = hu.q(w)
nch = Cin // 32
for z in range(S):
    lo, hi = z * nch // S * 32, (z + 1) * nch // S * 32
    ref = U.conv2d_fwd(x[..., lo:hi], wq[:, :, lo:hi, :], None, relu=False)   # [N][Ho][Wo][Cout]
    got = np.full((Ho, Wo, Cout), np.nan, np.float32)
    for slot in range(nslots):
        t, cob = slot // ncob, slot % ncob
        for wave in range(8):
            wco, wpx = wave // WPX, wave % WPX
            for e in range(NST):
                pt, pp = e // (CT // 2), e % (CT // 2)
                for lane in range(64):
                    g4, l15 = lane >> 4, lane & 15
                    ml = (wpx * PT + pt) * 16 + l15
                    yy, xx = t * TR + ml // SW, ml % SW
                    co = cob * 128 + (wco * (CT // 2) + pp) * 32 + 8 * g4
                    if yy < Ho and xx < Wo:
                        got[yy, xx, co:co + 4] = slab[z, slot, wave, e, 0, lane]
                        got[yy, xx, co + 4:co + 8] = slab[z, slot, wave, e, 1, lane]
    err = np.abs(got - ref[0])
    print("slice %d channels %d..%d: nan %d, max err %.3e (max|ref| %.3f); per tile max err %s; per cob %s" % (
        z, lo, hi, int(np.isnan(got).sum()), np.nanmax(err), np.abs(ref).max(),
        [float(np.nanmax(err[t * TR:(t + 1) * TR])) for t in range(ntile)], [float(np.nanmax(err[:, :, c * 128:(c + 1) * 128])) for c in range(ncob)]))
full = U.conv2d_fwd(x, wq, None, relu=False)
yh = hu.host(y)
print("finish: max err vs oracle %.3e" % np.abs(yh - hu.q(full)).max())
err = np.abs(yh - hu.q(full))[0]
tol = 0.02 * np.abs(full[0]) + 1e-3
bad = err > tol
print("bad fraction %.3f" % bad.mean())
print("bad by row   :", [round(float(bad[r].mean()), 2) for r in range(Ho)])
print("bad by column:", [round(float(bad[:, c].mean()), 2) for c in range(Wo)])
print("bad by 8-channel group (first 16):", [round(float(bad[:, :, g * 8:(g + 1) * 8].mean()), 2) for g in range(16)])
print("bad by channel within a group:", [round(float(bad[:, :, j::8].mean()), 2) for j in range(8)])
# is y some other slice combination?
part = [U.conv2d_fwd(x[..., z * 128:(z + 1) * 128], wq[:, :, z * 128:(z + 1) * 128, :], None, relu=False)[0] for z in range(4)]
for combo in ([0], [0, 1], [0, 1, 2], [1, 2, 3], [3]):
    s_ = sum(part[i] for i in combo)
    print("y vs sum of slices", combo, "max err %.3e" % np.abs(yh[0] - hu.q(s_)).max())
