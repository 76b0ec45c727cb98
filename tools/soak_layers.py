#!/usr/bin/env python3
"""Repeatability soak of every MFMA launch shape of a configuration: each op runs --reps times on the same inputs and every
result must be bit-identical to the first (a rare race or hazard shows up as a handful of differing elements).
usage: python tools/soak_layers.py [--L 5 --root 64 --P 388 --B 4] [--reps 300] [--ops fwd,bwd,wg,tfwd,tbwd,twg] [--only name]"""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from road_segmentation_unet_amd._lib import RsuSrc, call, lib  # noqa: E402
from bench_layers import layers, ptr, DEV  # noqa: E402


def convt_layers(L, root, P):
    hs = [(n, h) for n, h, _, _, _ in layers(L, root, P) if n.endswith("conv2")]
    out, nf = [], root * 2 ** (L - 1)
    h = hs[L - 1][1] - 2
    for i in range(L - 1):
        nf //= 2
        out.append(("up_%d" % i, h, 2 * nf, nf))
        h = 2 * h - 4
    return out


def soak(name, fn, outs, reps):
    fn()
    torch.cuda.synchronize()
    ref = [o.clone() for o in outs]
    bad, first = 0, ""
    for r in range(reps):
        for o in outs:
            o.fill_(7.0)
        fn()
        torch.cuda.synchronize()
        for o, g in zip(outs, ref):
            if not torch.equal(o, g):
                bad += 1
                if not first:
                    idx = torch.nonzero(o.float() != g.float())
                    first = "rep %d: %d elements differ, first %s last %s max %.4g" % (
                        r, idx.shape[0], idx[:6].tolist(), idx[-1].tolist(), float((o.float() - g.float()).abs().max()))
                    soak.last = (o.clone(), g.clone())
                break
    print("%-22s %s %d/%d bad %s" % (name, "OK " if bad == 0 else "BAD", bad, reps, first), flush=True)
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--L", type=int, default=5)
    ap.add_argument("--root", type=int, default=64)
    ap.add_argument("--P", type=int, default=388)
    ap.add_argument("--B", type=int, default=4)
    ap.add_argument("--reps", type=int, default=300)
    ap.add_argument("--ops", default="fwd,bwd,wg,tfwd,tbwd,twg")
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    ops = args.ops.split(",")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    B, tot = args.B, 0
    for name, hin, cin, cout, dil in layers(args.L, args.root, args.P):
        if args.only and args.only not in name:
            continue
        ho = hin - 2 * dil
        x = torch.randn((B, hin, hin, cin), device=DEV).to(torch.bfloat16)
        dz = torch.randn((B, ho, ho, cout), device=DEV).to(torch.bfloat16)
        y = torch.zeros((B, ho, ho, cout), device=DEV, dtype=torch.bfloat16)
        dx = torch.zeros_like(x)
        w = torch.randn((3, 3, cin, cout), device=DEV) * 0.05
        bias = torch.randn(cout, device=DEV)
        seg = (ctypes.c_int * 1)(cin)
        pf = torch.zeros(lib().rsu_packed_bytes(9, cout, seg, 1) // 2, dtype=torch.bfloat16, device=DEV)
        seg2 = (ctypes.c_int * 1)(cout)
        pb = torch.zeros(lib().rsu_packed_bytes(9, cin, seg2, 1) // 2, dtype=torch.bfloat16, device=DEV)
        call("rsu_pack_conv_fwd", ptr(w), ptr(pf), 3, cin, cout, seg, 1, st)
        call("rsu_pack_conv_bwd", ptr(w), ptr(pb), 3, cin, 0, cin, cout, st)
        dw, db = torch.zeros_like(w), torch.zeros(cout, device=DEV)
        ws = torch.zeros(lib().rsu_conv2d_bwd_weight_ws_floats(cin, cin, cout), device=DEV)
        src = RsuSrc(x.data_ptr(), hin, hin, cin, 0, 0)
        arr = (RsuSrc * 1)(src)
        tag = "%s H%d %d->%d" % (name, hin, cin, cout)
        if "fwd" in ops:
            tot += soak(tag + " fwd", lambda: call("rsu_conv2d_fwd", arr, 1, ptr(pf), ptr(bias), ptr(y), B, hin, hin, cout, dil, 1, 0, st), [y], args.reps)
        if "bwd" in ops:
            tot += soak(tag + " bwd", lambda: call("rsu_conv2d_bwd_data", ptr(dz), ptr(pb), ptr(dx), ptr(x), 0, B, hin, hin, cin, 0, cin, cout, dil, 0, st), [dx], args.reps)
        if "wg" in ops:
            tot += soak(tag + " wg", lambda: call("rsu_conv2d_bwd_weight", ctypes.byref(src), ptr(dz), ptr(dw), ptr(db), ptr(ws), B, ho, ho, cin, 0, cout, dil, 0, st), [dw, db], args.reps)
    for name, h, cin, cout in convt_layers(args.L, args.root, args.P):
        if args.only and args.only not in name:
            continue
        x = torch.randn((B, h, h, cin), device=DEV).to(torch.bfloat16)
        dy = torch.randn((B, 2 * h, 2 * h, cout), device=DEV).to(torch.bfloat16)
        y, dx = torch.zeros_like(dy), torch.zeros_like(x)
        K = torch.randn((2, 2, cout, cin), device=DEV) * 0.05
        b = torch.randn(cout, device=DEV)
        pf = torch.zeros(4 * lib().rsu_packed_bytes(1, cout, (ctypes.c_int * 1)(cin), 1) // 2, dtype=torch.bfloat16, device=DEV)
        pb = torch.zeros(lib().rsu_packed_bytes(4, cin, (ctypes.c_int * 1)(cout), 1) // 2, dtype=torch.bfloat16, device=DEV)
        call("rsu_pack_convT_fwd", ptr(K), ptr(pf), cin, cout, st)
        call("rsu_pack_convT_bwd", ptr(K), ptr(pb), cin, cout, st)
        dK, db = torch.zeros_like(K), torch.zeros(cout, device=DEV)
        ws = torch.zeros(lib().rsu_convT2x2_bwd_weight_ws_floats(cin, cout), device=DEV)
        tag = "%s H%d %d->%d" % (name, h, cin, cout)
        if "tfwd" in ops:
            tot += soak(tag + " tfwd", lambda: call("rsu_convT2x2_fwd", ptr(x), ptr(pf), ptr(b), ptr(y), B, h, h, cin, cout, 0, st), [y], args.reps)
        if "tbwd" in ops:
            tot += soak(tag + " tbwd", lambda: call("rsu_convT2x2_bwd_data", ptr(dy), ptr(pb), ptr(dx), ptr(x), 1.0, B, h, h, cin, cout, 0, st), [dx], args.reps)
        if "twg" in ops:
            tot += soak(tag + " twg", lambda: call("rsu_convT2x2_bwd_weight", ptr(x), ptr(dy), ptr(dK), ptr(db), ptr(ws), B, h, h, cin, cout, 0, st), [dK, db], args.reps)
    print("total bad launches:", tot)
    sys.exit(1 if tot else 0)


if __name__ == "__main__":
    main()
