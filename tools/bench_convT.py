"""Per-level timing of the transposed convolutions (forward, backward-data, weight gradient) at the bench configuration.
usage: python tools/bench_convT.py [--L 5 --root 64 --P 388 --B 4] [--gens 1,2]   (RSU_CT_GEN: 1 = igemm_fwd2's tap launches, 2 = igemm_ct)"""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from road_segmentation_unet_amd._lib import call, lib  # noqa: E402
from road_segmentation_unet_amd.unet import input_size_needed  # noqa: E402

DEV = "cuda:0"


def ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--L", type=int, default=5)
    ap.add_argument("--root", type=int, default=64)
    ap.add_argument("--P", type=int, default=388)
    ap.add_argument("--B", type=int, default=4)
    ap.add_argument("--gens", default="1,2")
    a = ap.parse_args()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    call("rsu_set_autotune", 2)   # developer tool: measure tile shapes at first sight (RSU_TUNE_MEASURE)
    h = input_size_needed(a.P, a.L)
    for _ in range(a.L - 1):
        h = (h - 4) // 2
    h -= 4
    nf = a.root * 2 ** (a.L - 1)
    tot = {}
    for i in range(a.L - 1):
        cin, cout = nf, nf // 2
        x = torch.randn((a.B, h, h, cin), device=DEV).to(torch.bfloat16)
        K = torch.randn((2, 2, cout, cin), device=DEV) * 0.05
        b = torch.zeros(cout, device=DEV)
        seg, seg2 = (ctypes.c_int * 1)(cin), (ctypes.c_int * 1)(cout)
        pf = torch.zeros(4 * lib().rsu_packed_bytes(1, cout, seg, 1) // 2, dtype=torch.bfloat16, device=DEV)
        pb = torch.zeros(lib().rsu_packed_bytes(4, cin, seg2, 1) // 2, dtype=torch.bfloat16, device=DEV)
        call("rsu_pack_convT_fwd", ptr(K), ptr(pf), cin, cout, st)
        call("rsu_pack_convT_bwd", ptr(K), ptr(pb), cin, cout, st)
        y = torch.zeros((a.B, 2 * h, 2 * h, cout), dtype=torch.bfloat16, device=DEV)
        dy = torch.randn((a.B, 2 * h, 2 * h, cout), device=DEV).to(torch.bfloat16)
        dx = torch.zeros_like(x)
        dK = torch.zeros_like(K)
        db = torch.zeros(cout, device=DEV)
        ws = torch.zeros(lib().rsu_convT2x2_bwd_weight_ws_floats(cin, cout), device=DEV)
        fl = 2.0 * a.B * h * h * cin * cout * 4
        line = "up_conv_%d  H %3d C %4d->%4d %6.1f GF |" % (i, h, cin, cout, fl / 1e9)
        for gen in a.gens.split(","):
            os.environ["RSU_CT_GEN"] = gen
            for op, fn in [("fwd", lambda: call("rsu_convT2x2_fwd", ptr(x), ptr(pf), ptr(b), ptr(y), a.B, h, h, cin, cout, 0, st)),
                           ("bwd", lambda: call("rsu_convT2x2_bwd_data", ptr(dy), ptr(pb), ptr(dx), ptr(x), 1.0, a.B, h, h, cin, cout, 0, st)),
                           ("wg", lambda: call("rsu_convT2x2_bwd_weight", ptr(x), ptr(dy), ptr(dK), ptr(db), ptr(ws), a.B, h, h, cin, cout, 0, st))]:
                if op == "wg" and gen != a.gens.split(",")[0]:
                    continue
                t = timeit(fn)
                line += " %sg%s %5.0fus %4.0fTF |" % (op, gen, t * 1e6, fl / t / 1e12)
                tot[(op, gen)] = tot.get((op, gen), 0.0) + t
        print(line, flush=True)
        os.environ.pop("RSU_CT_GEN", None)
        h, nf = 2 * h - 4, nf // 2
    print("totals (us):", {k: round(v * 1e6, 1) for k, v in tot.items()})


if __name__ == "__main__":
    main()
