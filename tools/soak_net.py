#!/usr/bin/env python3
"""Repeatability soak of the whole network: the same forward (and backward) pass runs --reps times on the same inputs and
every activation, activation gradient and weight gradient must be bit-identical to the first pass. Between two launches of
one kernel the other kernels of the net run, so instruction caches, LDS contents and clocks differ from launch to launch --
the conditions under which a rare register hazard shows (tools/soak_layers.py, one kernel back to back, does not provoke it).
usage: python tools/soak_net.py [--reps 400] [--fwd-only] [--L 5 --root 64 --P 388 --B 4] [--keep 1.0] [--dilated]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from road_segmentation_unet_amd.unet import UNet  # noqa: E402


def soak(reps, bwd=True, L=5, root=64, P=388, B=4, keep=1.0, max_report=5, verbose=True, dilated=False):
    """returns the number of passes that differed from the first one"""
    net = UNet(L, root, dilated, B, P, seed=2018, training=True)
    g = torch.Generator(device="cpu").manual_seed(7)
    net.x.copy_(torch.rand((B, net.S, net.S, 3), generator=g))
    net.labels.copy_((torch.rand((B, P, P), generator=g) < 0.2).to(torch.int64))
    scale = 1.0 / (B * P * P)

    def one():
        net.forward_device(keep=keep)
        if bwd:
            net.backward_device(scale)
        torch.cuda.synchronize()

    one()
    ref_act = {k: v.clone() for k, v in net.act.items()}
    ref_grad = {k: v.clone() for k, v in net.grad.items()} if bwd else {}
    ref_dw = net.flat_g.clone()
    nbad = 0
    for r in range(reps):
        one()
        bad = [k for k in ref_act if not torch.equal(net.act[k], ref_act[k])]
        badg = [k for k in ref_grad if not torch.equal(net.grad[k], ref_grad[k])]
        baddw = bwd and not torch.equal(ref_dw, net.flat_g)
        if bad or badg or baddw:
            nbad += 1
            if verbose and nbad <= max_report:
                print("pass", r, "activations", bad[:4], "gradients", badg[:4], "dW", baddw, flush=True)
                for k, cur, ref in [(k, net.act[k], ref_act[k]) for k in bad[:1]] + [(k, net.grad[k], ref_grad[k]) for k in badg[:1]]:
                    idx = torch.nonzero(cur.float() != ref.float())
                    print("   %s %s: %d elements, first %s last %s" % (k, tuple(cur.shape), idx.shape[0], idx[0].tolist(), idx[-1].tolist()))
                    for i in idx[:4].tolist():
                        print("      %s got %.6e want %.6e" % (i, float(cur[tuple(i)]), float(ref[tuple(i)])))
    if verbose:
        print("passes %d (%s) differing %d" % (reps, "fwd+bwd" if bwd else "fwd", nbad))
    return nbad


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=400)
    ap.add_argument("--fwd-only", action="store_true")
    ap.add_argument("--L", type=int, default=5)
    ap.add_argument("--root", type=int, default=64)
    ap.add_argument("--P", type=int, default=388)
    ap.add_argument("--B", type=int, default=4)
    ap.add_argument("--keep", type=float, default=1.0)
    ap.add_argument("--dilated", action="store_true")
    a = ap.parse_args()
    sys.exit(1 if soak(a.reps, not a.fwd_only, a.L, a.root, a.P, a.B, a.keep, dilated=a.dilated) else 0)
