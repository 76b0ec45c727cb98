#!/usr/bin/env python3
"""Probe (GPU box): can the Momentum pass of step t run BESIDE the forward pass of step t+1? A persistent MFMA kernel needs empty CUs
(profiles/r04/defer_first_probe.txt), so the update is confined to a few CUs with hipExtStreamCreateWithCUMask and the forward pass plans
for the rest. Times, for the c3 / c2 models: forward alone (256 and 192 CUs), update alone (whole chip / masked stream), both together.
usage: probe_cumask.py [c3|c2] [cus_for_update=64]"""
import ctypes, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from road_segmentation_unet_amd import _lib
from road_segmentation_unet_amd._lib import call
from road_segmentation_unet_amd.unet import UNet, input_size_needed
which = sys.argv[1] if len(sys.argv) > 1 else "c3"
ncu_up = int(sys.argv[2]) if len(sys.argv) > 2 else 64
L, dil, B = (6, True, 1) if which == "c3" else (5, False, 4)
m = UNet(L, 64, dil, B, 388, training=True)
S = input_size_needed(388, L)
m.x.copy_(torch.rand(B, S, S, 3)); m.labels.copy_((torch.rand(B, 388, 388) < 0.2).to(torch.int64))
m.ensure_tuned()
inv = 1.0 / (B * 388 * 388)
for _ in range(3):
    m.forward_device(); m.backward_device(inv); m.apply_momentum(0.01, 0.9)
torch.cuda.synchronize()
h = _lib._hip_runtime()
h.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
bits = [1 if i < ncu_up else 0 for i in range(256)]   # (a contiguous run of mask bits: a strided mask is accepted but confines nothing)
words = (ctypes.c_uint32 * 8)(*[sum(bits[w * 32 + b] << b for b in range(32)) for w in range(8)])
us = ctypes.c_void_p()
rc = h.hipExtStreamCreateWithCUMask(ctypes.byref(us), 8, words)
print("hipExtStreamCreateWithCUMask rc", rc, "stream", us.value, "CUs", sum(bits))
tab = m._update_table
def update(stream):
    call("rsu_update_table_run", ctypes.c_void_p(tab[0].data_ptr()), tab[1], tab[2], 0.0, 0.9, 1.0, stream)   # lr 0: weights stay
main = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def timeit(f, n=8):
    torch.cuda.synchronize(); ts = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(ts))
t_fwd = timeit(lambda: m.forward_device())
t_up = timeit(lambda: update(main))
t_upm = timeit(lambda: update(us))
call("rsu_set_cu_budget", 256 - ncu_up)
m._tuned = set(); m.ensure_tuned()
t_fwd_part = timeit(lambda: m.forward_device())
def both():
    update(us); m.forward_device()
t_both = timeit(both)
def both_full():
    update(main); m.forward_device()
t_serial = timeit(both_full)
print("%s: forward %.3f ms (256 CUs) / %.3f ms (planned for %d); update %.3f ms whole chip / %.3f ms on %d CUs; update(masked) beside forward(%d) %.3f ms; "
      "update then forward on one stream (planned for %d) %.3f ms" % (which, t_fwd, t_fwd_part, 256 - ncu_up, t_up, t_upm, ncu_up, 256 - ncu_up, t_both, 256 - ncu_up, t_serial))
