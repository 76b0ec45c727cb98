#!/usr/bin/env python3
"""Developer tool (GPU box): what this chip's HBM does for a pure write, a pure read and a copy (torch fill_ / sum / copy_ over 1 GiB, best of 5) -- the
ceilings the bandwidth-bound kernels of profiles/rNN/hbm_kernels.md should be read against (a kernel that is 80 % writes cannot reach the read rate)."""
import torch
D = "cuda:0"
n = 256 << 20
a = torch.empty(n, dtype=torch.float32, device=D)
b = torch.empty(n, dtype=torch.float32, device=D)


def best(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        t.append(e0.elapsed_time(e1) * 1e-3)
    return min(t)


tw = best(lambda: a.fill_(1.0))
tr = best(lambda: a.sum())
tc = best(lambda: b.copy_(a))
print("write only (fill_ 1 GiB): %.0f GB/s | read only (sum 1 GiB): %.0f GB/s | copy (1 GiB read + 1 GiB written): %.0f GB/s of traffic" % (
    4 * n / tw / 1e9, 4 * n / tr / 1e9, 8 * n / tc / 1e9))
