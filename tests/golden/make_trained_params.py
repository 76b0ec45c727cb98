"""Generates tests/golden/real_crops_trained_params.npz: weights of the L=3, root=32 U-Net trained (HIP path, MI355X) on the 12 training
crops of real_crops.npz until the held-out pixel-F1 passes 0.35 -- a fixed, reasonably trained network for the inference-parity test
(tests/test_gpu_parity_full.py). Training from scratch inside the test made its F1 depend on the summation order of the weight
gradients (0.38 / 0.15 / 0.0 after 240 steps for builds that differ in nothing else); the comparison HIP vs oracle needs weights, not a
particular way of getting them. Run on the GPU box:  python tests/golden/make_trained_params.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from road_segmentation_unet_amd.model import ConvolutionalModel, Options, pixel_f1  # noqa: E402

z = np.load(os.path.join(HERE, "real_crops.npz"))
x = z["x"].astype(np.float64) / 255.0
y = (z["y"].astype(np.float64) / 255.0 >= 0.5) * 1.0
P, L, root, B = int(z["P"]), 3, 32, 4
xtr, ytr, xte, yte = x[:12], y[:12], x[12:], y[12:]
m = ConvolutionalModel(Options(num_layers=L, root_size=root, patch_size=P, batch_size=B, dropout=1.0, lr=0.02, seed=21, logdir=None))
xtr13, ytr13 = np.concatenate([xtr, xtr[:1]]), np.concatenate([ytr, ytr[:1]])


def heldout_f1():
    m.net.training = False
    out = []
    for i in range(0, xte.shape[0], B):
        xb = xte[i:i + B]
        m.net.x.zero_()
        m.net.x[:xb.shape[0]].copy_(torch.from_numpy(xb.astype(np.float32)))
        m.net.forward_device()
        out.append(m.net.prob[:xb.shape[0]].cpu().numpy().copy())
    m.net.training = True
    return pixel_f1(np.concatenate(out), yte)


best = 0.0
for epoch in range(400):
    st = m.train(xtr13, ytr13, None, None)
    if epoch % 10 == 9:
        f = heldout_f1()
        print("\nepoch %d loss %.4f held-out F1 %.4f" % (epoch, st["loss"], f), flush=True)
        if f > 0.35 and epoch >= 59:
            best = f
            break
params = {k: v for k, v in m.net.state_dict().items() if not k.endswith("/Momentum") and k != "global_step"}
out = os.environ.get("OUT", os.path.join(HERE, "real_crops_trained_params.npz"))
np.savez_compressed(out, **{k.replace("/", "__"): np.asarray(v, np.float32) for k, v in params.items()})
print("saved %s (held-out F1 %.4f, %d arrays, %.2f MB)" % (out, best, len(params), os.path.getsize(out) / 1e6))
