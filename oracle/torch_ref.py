"""TEST / BASELINE INFRASTRUCTURE -- never imported by the product path (road_segmentation_unet_amd/).

The graph of the reference's src/unet.py:12-97 and the loss / optimizer of src/tf_aerial_images.py:103-122 in STOCK PyTorch CPU
ops (F.conv2d / conv_transpose2d / max_pool2d / cross_entropy + autograd). Two users:
  * tests/test_oracle_vs_torch.py: float64, the independent second opinion for oracle/unet_oracle.c;
  * bench.py `cpu_baseline_torch`: float32 through oneDNN on the host cores -- the CPU stand-in BASELINE.md section 3 promised
    beside the C port (TensorFlow 1.4 itself cannot be installed; kind "stand-in").
Not the reference itself: parity stays unpinned at the TensorFlow boundary (oracle/unet_oracle.py header)."""
import time

import numpy as np
import torch
import torch.nn.functional as F


def torch_unet(params, X, L, root, dilated, dtype=torch.float64):
    """logits [N, 2, P, P] (NCHW) and the dict of leaf tensors (requires_grad) for `params` in the reference's names/layouts"""
    P = {k: torch.from_numpy(np.asarray(v)).to(dtype).requires_grad_(True) for k, v in params.items()}

    def conv(x, name, dil=1, relu=True):
        w = P[name + "/kernel"].permute(3, 2, 0, 1)
        y = F.conv2d(x, w, P[name + "/bias"], dilation=dil)
        return F.relu(y) if relu else y

    x = torch.from_numpy(np.asarray(X)).to(dtype).permute(0, 3, 1, 2).contiguous()
    net = conv(x - 0.5, "color_space_adjust", relu=False)
    skips = []
    for i in range(L):
        dil = None
        if dilated:
            dil = conv(conv(net, "conv_dilut_%d/atrous_conv1" % i, 2), "conv_dilut_%d/atrous_conv2" % i, 2)
        net = conv(conv(net, "conv_%d/conv1" % i), "conv_%d/conv2" % i)
        skips.append((net, dil))
        net = F.max_pool2d(net, 2, 2)
    net = skips.pop()[0]
    for i in range(L - 1):
        kt = P["up_conv_%d/kernel" % i].permute(3, 2, 0, 1)
        net = F.conv_transpose2d(net, kt, P["up_conv_%d/bias" % i], stride=2)
        s, d = skips.pop()
        h, w = net.shape[2], net.shape[3]

        def crop(t):
            oy, ox = (t.shape[2] - h) // 2, (t.shape[3] - w) // 2
            return t[:, :, oy:oy + h, ox:ox + w]
        parts = [crop(s)] + ([crop(d)] if dilated else []) + [net]
        net = torch.cat(parts, 1)
        net = conv(conv(net, "conv_%d/conv1" % (L + i)), "conv_%d/conv2" % (L + i))
    return conv(net, "weight_output", relu=False), P


def timed_train_step_fp32(params, X, labels, L, root, dilated, lr=0.01, momentum=0.9, threads=None, repeats=3):
    """forward + loss + backward + Momentum update in float32 on the CPU. One untimed warm-up step (oneDNN primitive creation,
    thread-pool start, first-touch of the buffers), then `repeats` timed steps on parameter tensors created OUTSIDE the timed
    region; returns (best seconds, loss of the last step)"""
    if threads:
        torch.set_num_threads(int(threads))
    acc = {k: torch.zeros(np.asarray(v).shape, dtype=torch.float32) for k, v in params.items()}
    cur = {k: np.asarray(v, dtype=np.float32).copy() for k, v in params.items()}
    lab = torch.from_numpy(np.asarray(labels)).reshape(-1)
    best, loss_v = float("inf"), 0.0
    for it in range(repeats + 1):
        logits, P = torch_unet(cur, X, L, root, dilated, dtype=torch.float32)   # (leaf creation: outside the timed region below)
        t0 = time.time()
        logits, P = torch_unet(cur, X, L, root, dilated, dtype=torch.float32)
        lt = logits.permute(0, 2, 3, 1).reshape(-1, 2)
        loss = F.cross_entropy(lt, lab)
        loss.backward()
        with torch.no_grad():
            for k, p in P.items():
                if p.grad is None:
                    continue
                acc[k].mul_(momentum).add_(p.grad)
                p.sub_(lr * acc[k])
        dt = time.time() - t0
        if it > 0:
            best = min(best, dt)
        loss_v = float(loss)
    return best, loss_v
