#!/usr/bin/env python3
"""Per-layer timing of the conv ops at the BASELINE config shapes (developer tool, runs on the GPU box).
usage: python tools/bench_layers.py [--L 5 --root 64 --P 388 --B 4] [--cfgs -1,0,1,2,3,4] [--ops fwd,bwd,wg] [--ncu 128] [--cold]"""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from road_segmentation_unet_amd._lib import RsuSrc, call, lib  # noqa: E402
from road_segmentation_unet_amd.unet import input_size_needed  # noqa: E402

DEV = "cuda:0"


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


_COLD = None   # --cold: a 512-MiB buffer written between repetitions, so that a layer's weights and inputs come from HBM as they do in the step


def timeit(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    if _COLD is not None:
        # every repetition behind a write that displaces L2 and the Infinity Cache (256 MB): the in-step condition (VERDICT r5 "What's weak" 5:
        # hot-cache tables overstate in-step rates by 13-47 %); timed one by one, the flush is outside the events
        tot = 0.0
        for i in range(reps):
            _COLD.fill_(float(i))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            tot += e0.elapsed_time(e1)
        return tot / reps * 1e-3
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def layers(L, root, P, dilated=False):
    """(name, Hin, Cin, Cout, dil) of every MFMA 3x3 conv, forward order"""
    S = input_size_needed(P, L)
    out = []
    h, nf, cin = S, root, 3
    for i in range(L):
        if i > 0:
            out.append(("conv_%d/conv1" % i, h, cin, nf, 1))
        out.append(("conv_%d/conv2" % i, h - 2, nf, nf, 1))
        if dilated and i < L - 1:
            if i > 0:
                out.append(("dil_%d/conv1" % i, h, cin, nf, 2))
            out.append(("dil_%d/conv2" % i, h - 4, nf, nf, 2))
        if i < L - 1:
            h = (h - 4) // 2
            cin, nf = nf, nf * 2
    h -= 4
    for i in range(L - 1):
        nf //= 2
        h *= 2
        out.append(("conv_%d/conv1" % (L + i), h, (3 if dilated else 2) * nf, nf, 1))
        out.append(("conv_%d/conv2" % (L + i), h - 2, nf, nf, 1))
        h -= 4
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--L", type=int, default=5)
    ap.add_argument("--root", type=int, default=64)
    ap.add_argument("--P", type=int, default=388)
    ap.add_argument("--B", type=int, default=4)
    ap.add_argument("--dilated", action="store_true")
    ap.add_argument("--cfgs", default="-1")
    ap.add_argument("--ops", default="fwd,bwd,wg")
    ap.add_argument("--only", default="")
    ap.add_argument("--gens", default="", help="comma list of RSU_FWD_GEN values to time side by side (fwd / bwd ops)")
    ap.add_argument("--dbg", default="", help="comma list of RSU_FWD_DBG values to time side by side")
    ap.add_argument("--ncu", type=int, default=0, help="CU budget the launches plan for (0: the library default, 256)")
    ap.add_argument("--cold", action="store_true", help="write 512 MiB between repetitions: every launch starts with cold L2 / Infinity Cache, as in the step")
    args = ap.parse_args()
    if args.cold:
        global _COLD
        _COLD = torch.empty(128 << 20, dtype=torch.float32, device=DEV)
    cfgs = [int(c) for c in args.cfgs.split(",")]
    gens = [g for g in args.gens.split(",") if g] or [None]
    dbgs = [g for g in args.dbg.split(",") if g] or [None]
    ops = args.ops.split(",")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    call("rsu_set_autotune", 2)   # developer tool: measure tile shapes at first sight (RSU_TUNE_MEASURE)
    nk = int(lib().rsu_conv_splitk_ws_floats())
    kws = torch.zeros(nk, device=DEV)   # split-K workspace (rsu_conv2d_*_k): the launches may cut their reductions as the product's do
    ncu = args.ncu
    tot = {}
    for name, hin, cin, cout, dil in layers(args.L, args.root, args.P, args.dilated):
        if args.only and args.only not in name:
            continue
        B = args.B
        ho = hin - 2 * dil
        x = torch.randn((B, hin, hin, cin), device=DEV).to(torch.bfloat16)
        dz = torch.randn((B, ho, ho, cout), device=DEV).to(torch.bfloat16)
        y = torch.zeros((B, ho, ho, cout), device=DEV, dtype=torch.bfloat16)
        dx = torch.zeros_like(x)
        w = torch.randn((3, 3, cin, cout), device=DEV) * 0.05
        bias = torch.zeros(cout, device=DEV)
        seg = (ctypes.c_int * 1)(cin)
        pf = torch.zeros(lib().rsu_packed_bytes(9, cout, seg, 1) // 2, dtype=torch.bfloat16, device=DEV)
        seg2 = (ctypes.c_int * 1)(cout)
        pb = torch.zeros(lib().rsu_packed_bytes(9, cin, seg2, 1) // 2, dtype=torch.bfloat16, device=DEV)
        call("rsu_pack_conv_fwd", ptr(w), ptr(pf), 3, cin, cout, seg, 1, st)
        call("rsu_pack_conv_bwd", ptr(w), ptr(pb), 3, cin, 0, cin, cout, st)
        dw = torch.zeros_like(w)
        db = torch.zeros(cout, device=DEV)
        ws = torch.zeros(lib().rsu_conv2d_bwd_weight_ws_floats(cin, cin, cout), device=DEV)
        src = RsuSrc(x.data_ptr(), hin, hin, cin, 0, 0)
        arr = (RsuSrc * 1)(src)
        fl = 2.0 * B * ho * ho * cout * cin * 9
        line = "%-14s H%4d C%4d->%4d d%d %7.1f GF |" % (name, hin, cin, cout, dil, fl / 1e9)
        for op in ops:
            for cfg, gen, dbg in [(c, g, d) for c in (cfgs if not op.startswith("wg") else [-1]) for g in (gens if not op.startswith("wg") else [None]) for d in (dbgs if not op.startswith("wg") else [None])]:
                if gen is not None:
                    os.environ["RSU_FWD_GEN"] = gen
                if dbg is not None:
                    os.environ["RSU_FWD_DBG"] = dbg
                if cfg >= 0:
                    os.environ["RSU_FWD2_CFG"] = str(cfg)
                else:
                    os.environ.pop("RSU_FWD2_CFG", None)
                try:
                    if op == "fwd":
                        t = timeit(lambda: call("rsu_conv2d_fwd_k", arr, 1, ptr(pf), ptr(bias), ptr(y), B, hin, hin, cout, dil, 1, ncu, ptr(kws), nk, st))
                    elif op == "bwd":
                        t = timeit(lambda: call("rsu_conv2d_bwd_data_k", ptr(dz), ptr(pb), ptr(dx), ptr(x), 0, B, hin, hin, cin, 0, cin, cout, dil, ncu, ptr(kws), nk, st))
                    elif op == "bwdnm":  # backward-data without the ReLU mask (A/B: cost of the mask loads in the epilogue)
                        t = timeit(lambda: call("rsu_conv2d_bwd_data_k", ptr(dz), ptr(pb), ptr(dx), None, 0, B, hin, hin, cin, 0, cin, cout, dil, ncu, ptr(kws), nk, st))
                    else:  # "wg", or "wg1" / "wg2" = igemm_wgrad / igemm_wgpp (RSU_WG_GEN)
                        if len(op) > 2:  # "wg2d3" = generation 2 with RSU_WG_DBG=3 (timing ablation)
                            g_, _, d_ = op[2:].partition("d")
                            os.environ["RSU_WG_GEN"] = g_
                            os.environ["RSU_WG_DBG"] = d_ or "0"
                        t = timeit(lambda: call("rsu_conv2d_bwd_weight", ctypes.byref(src), ptr(dz), ptr(dw), ptr(db), ptr(ws), B, ho, ho, cin, 0, cout, dil, ncu, st))
                    tag = ("" if cfg < 0 else "[%d]" % cfg) + ("" if gen is None else "g" + gen) + ("" if dbg is None else "d" + dbg)
                    line += " %s%s %6.0fus %5.0fTF |" % (op, tag, t * 1e6, fl / t / 1e12)
                    tot[(op, tag)] = tot.get((op, tag), 0.0) + t
                except Exception as ex:
                    line += " %s[%d] n/a |" % (op, cfg)
        os.environ.pop("RSU_FWD2_CFG", None)
        os.environ.pop("RSU_FWD_GEN", None)
        os.environ.pop("RSU_FWD_DBG", None)
        os.environ.pop("RSU_WG_GEN", None)
        os.environ.pop("RSU_WG_DBG", None)
        print(line, flush=True)
    print("totals (ms):", {k: round(v * 1e3, 3) for k, v in tot.items()})


if __name__ == "__main__":
    main()
