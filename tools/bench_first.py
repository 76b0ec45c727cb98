#!/usr/bin/env python3
"""Developer tool (GPU box): level 0's entry at the c2 shape (B = 4, 572 px, 64 channels): rsu_color_adjust_fwd + rsu_conv_first_fwd (rounds 1-5)
against rsu_color_conv_first_fwd (round 6: one launch from the f32 input); achieved GB/s over the algorithmic bytes.
usage: python tools/bench_first.py [B=4] [S=572] [Cout=64]   (RSU_LIB_PATH selects the build)"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from road_segmentation_unet_amd._lib import call, lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
S = int(sys.argv[2]) if len(sys.argv) > 2 else 572
C = int(sys.argv[3]) if len(sys.argv) > 3 else 64
D = "cuda:0"
ptr = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
x = torch.rand((B, S, S, 3), device=D)
w0, b0 = torch.randn((3, 3), device=D) * 0.5, torch.randn(3, device=D) * 0.1
w1, b1 = torch.randn((3, 3, 3, C), device=D) * 0.3, torch.randn(C, device=D) * 0.1
in16 = torch.zeros((B, S, S, 16), dtype=torch.bfloat16, device=D)
y = torch.zeros((B, S - 2, S - 2, C), dtype=torch.bfloat16, device=D)
pk = torch.zeros(lib().rsu_packed_first_bytes(C) // 2, dtype=torch.bfloat16, device=D)
call("rsu_pack_conv_first", ptr(w1), ptr(pk), C, st)
flush = torch.empty(128 << 20, dtype=torch.float32, device=D)


def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    tot = 0.0
    for i in range(reps):
        flush.fill_(float(i))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / reps * 1e-3


def two():
    call("rsu_color_adjust_fwd", ptr(x), ptr(w0), ptr(b0), ptr(in16), B * S * S, 1.0, 0, st)
    call("rsu_conv_first_fwd", ptr(in16), ptr(pk), ptr(b1), ptr(y), B, S, S, C, 1, 0, st)


def conv_only():
    call("rsu_conv_first_fwd", ptr(in16), ptr(pk), ptr(b1), ptr(y), B, S, S, C, 1, 0, st)


def one():
    call("rsu_color_conv_first_fwd", ptr(x), ptr(w0), ptr(b0), ptr(pk), ptr(b1), ptr(y), B, S, S, C, 1, 0, st)


npx, nout = B * S * S, B * (S - 2) * (S - 2)
for name, fn, byts in (("colour adjust + first conv, two launches", two, npx * (12 + 32 + 32) + nout * 2 * C),
                       ("first conv alone (in16 given)", conv_only, npx * 32 + nout * 2 * C),
                       ("colour adjust + first conv, one launch", one, npx * 12 + nout * 2 * C)):
    t = timeit(fn)
    print("%-44s %7.1f us  %6.1f MB  %5.0f GB/s (cold caches: a 512-MiB write in front of every repetition)" % (name, t * 1e6, byts / 1e6, byts / t / 1e9))
