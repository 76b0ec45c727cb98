"""Generates tests/golden/real388_trained_params.npz + tests/golden/real388_heldout.npz: an L=5, root=16 U-Net (patch_size 388, input 572:
the BASELINE geometry at a quarter of the width) trained by the HIP path (MI355X) on 96 of the reference's 100 real 400-px training
images until the pixel-F1 on the 4 held-out images passes 0.8, and those 4 held-out images with their ground truth. The inference-parity
test (tests/test_gpu_parity_full.py::test_real_388_patches_pixel_f1_within_1e3_of_the_float32_oracle) runs the held-out 388-patches
through these weights on the HIP path and on the float32 oracle.

The training images travel to the GPU box as tests/golden/_train388_data.npz (git-ignored, written in the build container from
/root/reference/data/training by the snippet in this file's history: images.load_train_data -> uint8); only the 4 held-out images and
the trained weights are committed. Run on the GPU box:  python tests/golden/make_trained_params_388.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from road_segmentation_unet_amd import images as dimages  # noqa: E402
from road_segmentation_unet_amd.model import ConvolutionalModel, Options, pixel_f1  # noqa: E402
from road_segmentation_unet_amd.unet import input_size_needed  # noqa: E402

z = np.load(os.path.join(HERE, "_train388_data.npz"))
x8, y8 = z["x"], z["y"]
L, root, P, B, stride = 5, 16, 388, 4, 12
S = input_size_needed(P, L)
held = [7, 36, 63, 90]                      # held-out images (fixed)
train = [i for i in range(x8.shape[0]) if i not in held]
x = x8.astype(np.float32) / 255.0
y = (y8.astype(np.float32) / 255.0 >= 0.5) * 1.0


def patches_of(idx, flips=False):
    """the reference's training patches (tf_aerial_images.py:401-418 with angle 0): mirror border by (S-P)/2, windows at `stride`"""
    xs, ys = [], []
    off = (S - P) // 2
    for i in idx:
        xp = np.pad(x[i], ((off, off), (off, off), (0, 0)), mode="symmetric")
        for ox in range(0, 400 - P + 1, stride):
            for oy in range(0, 400 - P + 1, stride):
                xs.append(xp[oy:oy + S, ox:ox + S])
                ys.append(y[i][oy:oy + P, ox:ox + P])
    return np.stack(xs), np.stack(ys)


xtr, ytr = patches_of(train)
xte, yte = patches_of(held)
print("train patches", xtr.shape, "held-out patches", xte.shape, flush=True)
m = ConvolutionalModel(Options(num_layers=L, root_size=root, patch_size=P, batch_size=B, dropout=1.0, lr=float(os.environ.get("LR", "0.02")),
                               seed=5, logdir=None))
rng = np.random.RandomState(3)


def heldout():
    m.net.training = False
    out = []
    for i in range(0, xte.shape[0], B):
        xb = xte[i:i + B]
        m.net.x.zero_()
        m.net.x[:xb.shape[0]].copy_(torch.from_numpy(xb))
        m.net.forward_device()
        out.append(m.net.prob[:xb.shape[0]].cpu().numpy().copy())
    m.net.training = True
    return pixel_f1(np.concatenate(out), yte)


best, target = 0.0, float(os.environ.get("TARGET_F1", "0.8"))
for epoch in range(int(os.environ.get("EPOCHS", "120"))):
    # D4 augmentation on the host (the reference's intended stochastic flips / rot90, tf_aerial_images.py:173-210): 8 symmetries
    k, fl = rng.randint(4), rng.randint(2)
    xa, ya = np.rot90(xtr, k, axes=(1, 2)), np.rot90(ytr, k, axes=(1, 2))
    if fl:
        xa, ya = xa[:, :, ::-1], ya[:, :, ::-1]
    st = m.train(np.ascontiguousarray(xa), np.ascontiguousarray(ya), None, None)
    if epoch % 5 == 4:
        f = heldout()
        print("\nepoch %d loss %.4f held-out F1 %.4f" % (epoch, st["loss"], f), flush=True)
        if f > best:
            best = f
            params = {k_: v for k_, v in m.net.state_dict().items() if not k_.endswith("/Momentum") and k_ != "global_step"}
        if f >= target and epoch >= 19:
            break
out = os.environ.get("OUT", os.path.join(HERE, "real388_trained_params.npz"))
np.savez_compressed(out, **{k_.replace("/", "__"): np.asarray(v, np.float32) for k_, v in params.items()})
np.savez_compressed(os.path.join(os.path.dirname(out), "real388_heldout.npz"), x=x8[held], y=y8[held], held=np.asarray(held), P=P, S=S, L=L, root=root)
print("saved %s (best held-out F1 %.4f, %.2f MB)" % (out, best, os.path.getsize(out) / 1e6))
