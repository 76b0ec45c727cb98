"""The training-patch pool and the input path of the train loop.

The reference cuts EVERY training patch out of the rotated, mirror-extended images up front (tf_aerial_images.py:404-418 ->
images.extract_patches, images.py:35-85): a float64 array of 28-34 GB at the README configuration, from which each step feeds
`patches[batch_indices]` (tf_aerial_images.py:233-239). Here the pool is an index: patch k of the reference's array is a window of
extended image k // pps^2 at (x, y) = ((r // pps) * stride, (r % pps) * stride), r = k % pps^2 (x outer, y inner) -- the same
patches in the same order, cut when a batch asks for them.

  PatchPool        host float32 images; gather() returns numpy batches (what ConvolutionalModel.train accepts next to the
                   reference's plain arrays); BatchUploader moves them through pinned memory on a copy stream, one batch ahead.
  DevicePatchPool  the extended images live in HBM (a few GB of 288); load_batch() cuts the windows on the GPU straight into the
                   network's input buffers: no host work, no PCIe traffic per step. Optionally applies the stochastic D4
                   augmentation the reference intended but never ran (stochastic_images_augmentation, tf_aerial_images.py:173-210:
                   the graph feeds the augmented tensor itself, so the subgraph is bypassed): per sample,
                   flip up-down / flip left-right / transpose with probability 1/2 each, then rot90 by a uniform k in 0..3, applied
                   alike to the input window and its label patch. Off by default (parity with the reference's actual behaviour).
"""
import numpy as np
import torch


class PatchPool:
    def __init__(self, extended_images, extended_labels, input_size, patch_size, stride):
        ei, el = np.asarray(extended_images), np.asarray(extended_labels)
        assert ei.ndim == 4 and el.ndim == 3 and ei.shape[0] == el.shape[0]
        assert ei.shape[1] == ei.shape[2] and el.shape[1] == el.shape[2], "Assume square images"
        self.S, self.P, self.stride = int(input_size), int(patch_size), int(stride)
        assert (ei.shape[1] - self.S) % self.stride == 0 and (el.shape[1] - self.P) % self.stride == 0, \
            "Stride sliding should cover the whole image"
        self.pps = (ei.shape[1] - self.S) // self.stride + 1
        assert self.pps == (el.shape[1] - self.P) // self.stride + 1, "image and label pools must tile alike"
        self.images = np.ascontiguousarray(ei, dtype=np.float32)
        self.labels = np.ascontiguousarray(el >= 0.5, dtype=np.uint8)  # tf_aerial_images.py:220 binarises at 0.5
        self.shape = (self.images.shape[0] * self.pps * self.pps, self.S, self.S, self.images.shape[-1])

    def __len__(self):
        return self.shape[0]

    def locate(self, k):
        """(image, x, y) of patch k in the reference's extract_patches order"""
        n, r = divmod(int(k), self.pps * self.pps)
        return n, (r // self.pps) * self.stride, (r % self.pps) * self.stride

    def gather(self, indices):
        """patches float32 [b, S, S, C] and labels float32 [b, P, P] in {0, 1} == patches[indices], labels_patches[indices]"""
        b = len(indices)
        x = np.empty((b, self.S, self.S, self.images.shape[-1]), np.float32)
        y = np.empty((b, self.P, self.P), np.float32)
        for j, k in enumerate(indices):
            n, x0, y0 = self.locate(k)
            x[j] = self.images[n, y0:y0 + self.S, x0:x0 + self.S]
            y[j] = self.labels[n, y0:y0 + self.P, x0:x0 + self.P]
        return x, y


def d4_draw(rng, count):
    """per sample (flip_ud, flip_lr, transpose, k): the draws of tf_aerial_images.py:183-201 (proba > 0.5; floor(U * 4))"""
    u = rng.random_sample((count, 4))
    return [(bool(a > 0.5), bool(b > 0.5), bool(c > 0.5), int(np.floor(d * 4))) for a, b, c, d in u]


def d4_apply(t, op):
    """t: [H, W(, C)] tensor (square); tf.image.flip_up_down / flip_left_right / transpose_image / rot90 (counter-clockwise)"""
    ud, lr, tr, k = op
    if ud:
        t = torch.flip(t, (0,))
    if lr:
        t = torch.flip(t, (1,))
    if tr:
        t = t.transpose(0, 1)
    if k:
        t = torch.rot90(t, k, (0, 1))
    return t


class DevicePatchPool(PatchPool):
    def __init__(self, extended_images, extended_labels, input_size, patch_size, stride, device, augment=False, seed=2017):
        super().__init__(extended_images, extended_labels, input_size, patch_size, stride)
        self.device = torch.device(device)
        self.dev_images = torch.from_numpy(self.images).to(self.device)
        self.dev_labels = torch.from_numpy(self.labels).to(self.device)
        self.augment = bool(augment)
        self._rng = np.random.RandomState(seed)

    def load_batch(self, indices, x_out, labels_out):
        """x_out f32 [b, S, S, C], labels_out int64 [b, P, P] (device tensors: the network's input buffers)"""
        ops = d4_draw(self._rng, len(indices)) if self.augment else None
        for j, k in enumerate(indices):
            n, x0, y0 = self.locate(k)
            xi = self.dev_images[n, y0:y0 + self.S, x0:x0 + self.S]
            li = self.dev_labels[n, y0:y0 + self.P, x0:x0 + self.P]
            if ops is not None:
                xi, li = d4_apply(xi, ops[j]), d4_apply(li, ops[j])
            x_out[j].copy_(xi)
            labels_out[j].copy_(li)
        return ops


class BatchUploader:
    """Host batches -> HBM through two pinned buffers and a copy stream, one batch ahead of the step that consumes it: while
    step i runs, batch i+1 is converted to float32 into pinned memory and copied (20.5 MB at config 2: ~0.35 ms of PCIe Gen5)."""

    def __init__(self, net):
        self.net = net
        dev = net.device
        self.stream = torch.cuda.Stream(device=dev)
        self.pin_x = [torch.empty(net.x.shape, dtype=torch.float32).pin_memory() for _ in range(2)]
        self.pin_l = [torch.empty(net.labels.shape, dtype=torch.int64).pin_memory() for _ in range(2)]
        self.dev_x = [torch.empty_like(net.x) for _ in range(2)]
        self.dev_l = [torch.empty_like(net.labels) for _ in range(2)]
        self.ready = [torch.cuda.Event() for _ in range(2)]
        self.consumed = [torch.cuda.Event() for _ in range(2)]
        self._used = [False, False]

    def stage(self, slot, patches, labels):
        """start the upload of one batch (numpy or tensors; float64 inputs are narrowed here, on the host, once)"""
        if self._used[slot]:
            self.consumed[slot].synchronize()  # the step that read this slot's device copy has taken it
        np.copyto(self.pin_x[slot].numpy(), np.asarray(patches), casting="same_kind")
        np.copyto(self.pin_l[slot].numpy(), np.asarray(labels), casting="unsafe")
        with torch.cuda.stream(self.stream):
            self.dev_x[slot].copy_(self.pin_x[slot], non_blocking=True)
            self.dev_l[slot].copy_(self.pin_l[slot], non_blocking=True)
            self.ready[slot].record(self.stream)

    def commit(self, slot):
        """make the staged batch the network's input (device-to-device, on the compute stream)"""
        cur = torch.cuda.current_stream(self.net.device)
        cur.wait_event(self.ready[slot])
        self.net.x.copy_(self.dev_x[slot])
        self.net.labels.copy_(self.dev_l[slot])
        self.consumed[slot].record(cur)
        self._used[slot] = True
