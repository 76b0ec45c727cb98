#!/usr/bin/env python3
"""End-to-end ConvolutionalModel.train() rate at config 2 (num_layers=5 root_size=64 patch_size=388, batch 4) with the input path
included, next to bench.py's resident-input figure. Synthetic 400x400 images, rotations 0 and 45 degrees, the
reference's pipeline: expand_and_rotate -> patches -> shuffled epoch. Three input paths:
  arrays       the reference's float64 extract_patches arrays, through pool.BatchUploader (pinned, copy stream, one batch ahead)
  host_pool    pool.PatchPool: float32 rotated images on the host, patches cut per batch, same uploader
  device_pool  pool.DevicePatchPool: rotated images resident in HBM, patches cut on the GPU (the CLI default)
usage: python tools/bench_train_e2e.py [nimg=16] [epochs=2]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from road_segmentation_unet_amd import hostio  # noqa: E402
from road_segmentation_unet_amd.model import ConvolutionalModel, Options  # noqa: E402
from road_segmentation_unet_amd.pool import DevicePatchPool, PatchPool  # noqa: E402
from road_segmentation_unet_amd.unet import input_size_needed  # noqa: E402

nimg = int(sys.argv[1]) if len(sys.argv) > 1 else 16
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 2
L, root, P, B, stride = 5, 64, 388, 4, 12
S = input_size_needed(P, L)
off = (S - P) // 2
rng = np.random.RandomState(0)
imgs = rng.rand(nimg, 400, 400, 3).astype(np.float32)
gts = (rng.rand(nimg, 400, 400) < 0.2).astype(np.float32)
t0 = time.time()
ext = hostio.expand_and_rotate(imgs, [0, 45], off)
gte = hostio.expand_and_rotate(gts, [0, 45], 0)
t_rot = time.time() - t0
out = {"config": "num_layers=5 root_size=64 patch_size=388 batch=4, %d images x 2 angles, stride %d" % (nimg, stride), "rotate_s": t_rot}
for kind in ("arrays", "host_pool", "device_pool"):
    m = ConvolutionalModel(Options(num_layers=L, root_size=root, patch_size=P, batch_size=B, dropout=0.8, stride=stride, seed=1, logdir=None))
    t0 = time.time()
    if kind == "arrays":
        px = hostio.extract_patches(ext, patch_size=S, predict_patch_size=P, stride=stride)   # float64, like the reference
        py = hostio.extract_patches(gte, patch_size=P, stride=stride)
        pool_bytes = px.nbytes + py.nbytes
    elif kind == "host_pool":
        px, py = PatchPool(ext, gte, S, P, stride), None
        pool_bytes = px.images.nbytes + px.labels.nbytes
    else:
        px, py = DevicePatchPool(ext, gte, S, P, stride, device=m.net.device), None
        pool_bytes = px.images.nbytes + px.labels.nbytes
    t_pool = time.time() - t0
    m.train(px, py, None, None)   # first epoch: tile-shape tuning, allocations
    torch.cuda.synchronize()
    t0 = time.time()
    n = 0
    for _ in range(epochs):
        n += m.train(px, py, None, None)["patches"]
    torch.cuda.synchronize()
    dt = time.time() - t0
    out[kind] = {"patches_per_s": n / dt, "ms_per_step": dt / (n / B) * 1e3, "pool_build_s": t_pool, "pool_GB": pool_bytes / 1e9, "patches": n}
    del m, px, py
    torch.cuda.empty_cache()
print(json.dumps(out))
