import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from road_segmentation_unet_amd.unet import UNet, input_size_needed
for (L, root, P, B, dil) in [(2, 32, 100, 5, False), (3, 64, 148, 4, True), (5, 64, 388, 4, False)]:
    S = input_size_needed(P, L)
    g = torch.Generator().manual_seed(41)
    x = torch.rand((B, S, S, 3), generator=g)
    labels = (torch.rand((B, P, P), generator=g) < 0.2).to(torch.int64)
    grads = {}
    for spec in ("0", "128,128", "0", "128,128"):
        os.environ["RSU_SPLIT_CHIP"] = spec
        net = UNet(L, root, dil, B, P, seed=42, training=True)
        net.x.copy_(x); net.labels.copy_(labels)
        for it in range(2):
            net.forward_device(); net.backward_device(1.0 / (B * P * P))
        torch.cuda.synchronize()
        gr = net.flat_g[:net.n_live].clone()
        grads.setdefault(spec, []).append(gr)
        del net
    a, b = grads["0"][0], grads["128,128"][0]
    print(L, root, P, B, dil, "repeat0", torch.equal(grads["0"][0], grads["0"][1]), "repeat128", torch.equal(grads["128,128"][0], grads["128,128"][1]),
          "maxabs", float(a.abs().max()), "maxdiff", float((a - b).abs().max()), "rel", float((a - b).abs().max() / a.abs().max()))
    # per-variable worst
    net = UNet(L, root, dil, B, P, seed=42, training=True)
    worst = []
    for n, (lo, hi, s) in net._slices.items():
        if hi <= net.n_live and hi > lo:
            d = float((a[lo:hi] - b[lo:hi]).abs().max()); m = float(a[lo:hi].abs().max())
            worst.append((d / (m + 1e-30), n, d, m))
    worst.sort(reverse=True)
    for w in worst[:5]: print("   ", w)
    del net
