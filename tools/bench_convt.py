#!/usr/bin/env python3
"""Per-layer timing of the transposed-conv ops at the BASELINE config shapes (developer tool, GPU box).
usage: python tools/bench_convt.py [--L 5 --root 64 --P 388 --B 4]"""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from road_segmentation_unet_amd._lib import call, lib  # noqa: E402
from road_segmentation_unet_amd.unet import input_size_needed  # noqa: E402
from tools.bench_layers import ptr, timeit, DEV  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--L", type=int, default=5)
    ap.add_argument("--root", type=int, default=64)
    ap.add_argument("--P", type=int, default=388)
    ap.add_argument("--B", type=int, default=4)
    a = ap.parse_args()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    h = input_size_needed(a.P, a.L)
    for i in range(a.L - 1):
        h = (h - 4) // 2
    h -= 4
    nf = a.root * 2 ** (a.L - 1)
    tot = [0.0, 0.0, 0.0]
    for i in range(a.L - 1):
        cin, cout = nf, nf // 2
        B = a.B
        x = torch.randn((B, h, h, cin), device=DEV).to(torch.bfloat16)
        dy = torch.randn((B, 2 * h, 2 * h, cout), device=DEV).to(torch.bfloat16)
        y = torch.zeros_like(dy)
        dx = torch.zeros_like(x)
        K = torch.randn((2, 2, cout, cin), device=DEV) * 0.05
        b = torch.zeros(cout, device=DEV)
        one = (ctypes.c_int * 1)(cin)
        pf = torch.zeros(4 * lib().rsu_packed_bytes(1, cout, one, 1) // 2, dtype=torch.bfloat16, device=DEV)
        one2 = (ctypes.c_int * 1)(cout)
        pb = torch.zeros(lib().rsu_packed_bytes(4, cin, one2, 1) // 2, dtype=torch.bfloat16, device=DEV)
        call("rsu_pack_convT_fwd", ptr(K), ptr(pf), cin, cout, st)
        call("rsu_pack_convT_bwd", ptr(K), ptr(pb), cin, cout, st)
        dK = torch.zeros_like(K)
        ws = torch.zeros(lib().rsu_convT2x2_bwd_weight_ws_floats(cin, cout), device=DEV)
        fl = 2.0 * B * (2 * h) * (2 * h) * cin * cout
        t0 = timeit(lambda: call("rsu_convT2x2_fwd", ptr(x), ptr(pf), ptr(b), ptr(y), B, h, h, cin, cout, st))
        t1 = timeit(lambda: call("rsu_convT2x2_bwd_data", ptr(dy), ptr(pb), ptr(dx), ptr(x), 1.0, B, h, h, cin, cout, st))
        t2 = timeit(lambda: call("rsu_convT2x2_bwd_weight", ptr(x), ptr(dy), ptr(dK), None, ptr(ws), B, h, h, cin, cout, st))
        mb = (x.numel() + dy.numel()) * 2 / 1e6
        print("up_conv_%d H%4d C%4d->%4d %6.1f GF %6.1f MB | fwd %5.0fus %4.0fTF | bwd %5.0fus %4.0fTF | wg %5.0fus %4.0fTF" %
              (i, h, cin, cout, fl / 1e9, mb, t0 * 1e6, fl / t0 / 1e12, t1 * 1e6, fl / t1 / 1e12, t2 * 1e6, fl / t2 / 1e12), flush=True)
        for k, t in enumerate((t0, t1, t2)):
            tot[k] += t
        h = 2 * h - 4
        nf //= 2
    print("totals (ms): fwd %.3f bwd %.3f wg %.3f" % tuple(t * 1e3 for t in tot))


if __name__ == "__main__":
    main()
