"""CPU oracle (numpy) for the patch/stride tiler and the prediction ensemble of the reference.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
Pinned: every function here is checked against golden vectors produced by importing the reference's
own /root/reference/src/images.py in the build container (tests/golden/make_golden.py ->
tests/golden/tiler_golden.npz; test: tests/test_oracle_tiler_golden.py).

Each function cites the reference lines it restates. These are independent restatements (index
arithmetic / stride views), not copies of the reference loops.
"""
import numpy as np

FOREGROUND_THRESHOLD = 0.25  # src/constants.py:1
IMG_PATCH_SIZE = 16          # src/constants.py:2


def mirror_border(images, n):
    """images.py:269-281 -- np.pad 'symmetric' by n pixels on both spatial axes (edge pixel repeated)."""
    h, w = images.shape[1], images.shape[2]
    assert n <= h and n <= w
    iy = np.concatenate([np.arange(n - 1, -1, -1), np.arange(h), np.arange(h - 1, h - 1 - n, -1)]).astype(np.int64)
    ix = np.concatenate([np.arange(n - 1, -1, -1), np.arange(w), np.arange(w - 1, w - 1 - n, -1)]).astype(np.int64)
    return images[:, iy][:, :, ix]


def extract_patches(images, patch_size, stride=None, predict_patch_size=None):
    """images.py:35-85 -- square patches, image-major, then x (column) OUTER, y (row) INNER; float64 output."""
    if not predict_patch_size:
        predict_patch_size = patch_size
    assert (patch_size - predict_patch_size) % 2 == 0 and predict_patch_size <= patch_size
    if not stride:
        stride = patch_size
    n, h, w = images.shape[:3]
    assert h == w, "Assume square images"
    assert (h - patch_size) % stride == 0, "Stride sliding should cover the whole image"
    starts = np.arange(0, h - patch_size + 1, stride)
    pps = len(starts)
    win = np.lib.stride_tricks.sliding_window_view(images, (patch_size, patch_size), axis=(1, 2))
    # win: [n, h-p+1, w-p+1, (c,) p, p]
    win = win[:, starts][:, :, starts]  # [n, y, x, (c,) p, p]
    win = np.swapaxes(win, 1, 2)        # [n, x, y, ...]  (x outer)
    if images.ndim == 4:
        win = np.moveaxis(win, 3, -1)   # channel last
        out = win.reshape(n * pps * pps, patch_size, patch_size, images.shape[3])
    else:
        out = win.reshape(n * pps * pps, patch_size, patch_size)
    return np.ascontiguousarray(out, dtype=np.float64)


def images_from_patches(patches, stride=None):
    """images.py:131-164 -- overlap-add of [n_img, n_patches, p, p, c] in the same x-outer order, divided by hit count."""
    n_img, n_patches, p, _, c = patches.shape
    if stride is None:
        stride = p
    side = int(round(np.sqrt(n_patches)))
    assert side * side == n_patches, "Square image assumption broken"
    size = (side - 1) * stride + p
    acc = np.zeros((n_img, size, size, c), dtype=patches.dtype)
    hits = np.zeros((size, size), dtype=np.uint64)
    for k in range(n_patches):
        x0, y0 = (k // side) * stride, (k % side) * stride
        acc[:, y0:y0 + p, x0:x0 + p] += patches[:, k]
        hits[y0:y0 + p, x0:x0 + p] += 1
    return acc / hits[None, :, :, None]


def image_augmentation_ensemble(imgs):
    """images.py:376-396 -- [id, flip W, flip H, rot90 k=1,2,3 over axes (1,2)], float64, grouped by transform."""
    parts = [imgs, imgs[:, :, ::-1], imgs[:, ::-1]] + [np.rot90(imgs, k=k, axes=(1, 2)) for k in (1, 2, 3)]
    return np.concatenate(parts, axis=0).astype(np.float64)


def invert_image_augmentation_ensemble(masks):
    """images.py:399-417 -- inverse transforms, mean of 6. (The reference also mutates masks[:n] in place; the value
    returned is what is restated here.)"""
    assert masks.shape[0] % 6 == 0
    n = masks.shape[0] // 6
    g = [masks[i * n:(i + 1) * n] for i in range(6)]
    total = g[0] + g[1][:, :, ::-1] + g[2][:, ::-1]
    for i, k in enumerate((-1, -2, -3)):
        total = total + np.rot90(g[3 + i], k=k, axes=(1, 2))
    return total / 6


def labels_for_patches(patches):
    """images.py:88-99"""
    return (patches.mean(axis=(1, 2)) > FOREGROUND_THRESHOLD).astype(np.int64)


def quantize_mask(masks, threshold, patch_size):
    """images.py:256-266 -- per patch_size block: mean(mask >= 0.5) > threshold, broadcast back to the block."""
    n, size = masks.shape[0], masks.shape[1]
    out = masks.copy()
    nb = -(-size // patch_size)
    for by in range(nb):
        for bx in range(nb):
            blk = masks[:, by * patch_size:(by + 1) * patch_size, bx * patch_size:(bx + 1) * patch_size, 0]
            lab = (blk >= 0.5).reshape(n, -1).mean(axis=1) > threshold
            out[:, by * patch_size:(by + 1) * patch_size, bx * patch_size:(bx + 1) * patch_size, 0] = \
                lab[:, None, None]
    return out


def predictions_to_patches(predictions, patch_size):
    """images.py:167-180"""
    n = predictions.shape[0]
    return np.broadcast_to(np.resize(predictions, (n, 1, 1, 1)), (n, patch_size, patch_size, 1))


def submission_rows(masks, patch_size=IMG_PATCH_SIZE):
    """images.py:206-237 -- the CSV body of save_submission_csv as a list of strings (header excluded).
    Label grid index [j][i] comes from extract_patches' x-outer order, id is '{img:03d}_{16*j}_{16*i}'."""
    if masks.ndim == 4:
        masks = masks.squeeze(-1)
    n, h, w = masks.shape
    assert h == w, "images should be square"
    pps = h // patch_size
    labels = labels_for_patches(extract_patches(masks, patch_size)).reshape(n, pps, pps)
    rows = []
    for k in range(n):
        for j in range(pps):
            for i in range(pps):
                rows.append("{:03d}_{}_{},{}".format(k + 1, patch_size * j, patch_size * i, labels[k, j, i]))
    return rows


def predict_tiles_geometry(image_size, patch_size, input_size, stride):
    """Tile bookkeeping of ConvolutionalModel.predict (tf_aerial_images.py:288-293,316-320):
    offset, patches per side and the (x0, y0) origin of each tile in x-outer order."""
    offset = (input_size - patch_size) // 2
    assert (image_size - patch_size) % stride == 0, "Stride sliding should cover the whole image"
    pps = (image_size - patch_size) // stride + 1
    origins = [(xi * stride, yi * stride) for xi in range(pps) for yi in range(pps)]
    return offset, pps, origins
