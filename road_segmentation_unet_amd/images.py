"""Device-side patch/stride tiler and prediction ensemble, mirroring the hot-path subset of the reference's
src/images.py (same function names, argument meaning, ordering conventions and assertions).

Everything here runs on the GPU through librsu_hip.so (rsu_extract_tiles / rsu_overlap_add / rsu_overlap_finish) or is
pure index plumbing on torch tensors (flips / rot90). Inputs may be numpy arrays or torch tensors; results are torch
tensors on the device (call .cpu().numpy() for the reference's numpy convention).
"""
import ctypes

import numpy as np
import torch

from ._lib import call

DEV = "cuda:0"


def _dev(a, device=None):
    t = torch.as_tensor(a)
    return t.to(device or DEV, torch.float32).contiguous()


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def _stream(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def extract_mirrored_patches(images, patch_size, predict_patch_size, stride, t0=0, ntiles=None, out=None):
    """images.mirror_border(images, (patch_size - predict_patch_size) / 2) followed by
    images.extract_patches(..., patch_size, stride, predict_patch_size) (tf_aerial_images.py:288-293), fused on device.

    images: [n, H, H, 3] float. Returns float32 [ntiles, S, S, 3] for tile indices [t0, t0+ntiles) of the reference's
    ordering: image-major, then x (column) outer, y (row) inner (images.py:75-77). The padded image is never built."""
    images = _dev(images)
    n, H, W, C = images.shape
    assert H == W, "Assume square images"
    assert C == 3
    assert (patch_size - predict_patch_size) % 2 == 0 and predict_patch_size <= patch_size
    assert (H - predict_patch_size) % stride == 0, "Stride sliding should cover the whole image"
    pps = (H - predict_patch_size) // stride + 1
    total = n * pps * pps
    if ntiles is None:
        ntiles = total - t0
    if out is None:
        out = torch.empty((ntiles, patch_size, patch_size, 3), dtype=torch.float32, device=images.device)
    call("rsu_extract_tiles", _ptr(images), _ptr(out), n, H, patch_size, predict_patch_size, stride, t0, ntiles, _stream(images))
    return out


class OverlapAccumulator:
    """images.images_from_patches (images.py:131-164) as an accumulator: add() tile ranges in any split, finish() divides
    by the hit count. Deterministic (gather form, no atomics); partial accumulators of different ranks can be summed."""

    def __init__(self, num_images, image_size, patch_size, stride, device=DEV):
        assert (image_size - patch_size) % stride == 0, "Stride sliding should cover the whole image"
        self.n, self.H, self.P, self.stride = num_images, image_size, patch_size, stride
        self.acc = torch.zeros((num_images, image_size, image_size), dtype=torch.float32, device=device)
        self.hits = torch.zeros_like(self.acc)

    def add(self, probs, t0):
        """probs: [ntiles, P, P] float32 device tensor holding tiles t0 .. t0+ntiles-1"""
        probs = probs.contiguous()
        call("rsu_overlap_add", _ptr(probs), _ptr(self.acc), _ptr(self.hits), self.n, self.H, self.P, self.stride, t0, probs.shape[0],
             _stream(probs))

    def finish(self):
        out = torch.empty_like(self.acc)
        call("rsu_overlap_finish", _ptr(self.acc), _ptr(self.hits), _ptr(out), self.acc.numel(), _stream(out))
        return out.unsqueeze(-1)  # [n, H, H, 1] like the reference


def images_from_patches(patches, stride=None):
    """images.py:131-164 for [num_images, num_patches, p, p, 1] device/numpy input (single channel, the mask case)."""
    patches = _dev(patches)
    n, npatch, p, _, c = patches.shape
    assert c == 1
    if stride is None:
        stride = p
    side = int(round(np.sqrt(npatch)))
    assert side * side == npatch, "Square image assumption broken"
    acc = OverlapAccumulator(n, (side - 1) * stride + p, p, stride, device=patches.device)
    acc.add(patches.reshape(n * npatch, p, p), 0)
    return acc.finish()


def image_augmentation_ensemble(imgs):
    """images.py:376-396: [id, flip W, flip H, rot90 k=1,2,3 over axes (1,2)], grouped by transform."""
    imgs = torch.as_tensor(imgs)
    parts = [imgs, torch.flip(imgs, dims=(2,)), torch.flip(imgs, dims=(1,))] + [torch.rot90(imgs, k=k, dims=(1, 2)) for k in (1, 2, 3)]
    return torch.cat(parts, dim=0)


def invert_image_augmentation_ensemble(masks):
    """images.py:399-417: inverse transforms, mean of the 6 variants (returns a new tensor; the reference mutates its input)."""
    masks = torch.as_tensor(masks)
    assert masks.shape[0] % 6 == 0
    n = masks.shape[0] // 6
    g = [masks[i * n:(i + 1) * n] for i in range(6)]
    total = g[0] + torch.flip(g[1], dims=(2,)) + torch.flip(g[2], dims=(1,))
    for i, k in enumerate((-1, -2, -3)):
        total = total + torch.rot90(g[3 + i], k=k, dims=(1, 2))
    return total / 6
