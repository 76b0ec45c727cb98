"""Host-side data formats either side of the hot path (SURVEY.md section 8(f) "next" rows 2 and 3): PNG loading, the
offline rotation augmentation, mask quantisation and the Kaggle submission CSV. Plain numpy/scipy/PIL, off the timed path;
each function cites the reference lines it mirrors (/root/reference/src/images.py)."""
import glob
import os

import numpy as np

FOREGROUND_THRESHOLD = .25  # src/constants.py:1
IMG_PATCH_SIZE = 16         # src/constants.py:2
PIXEL_DEPTH = 255           # src/constants.py:5


def load(directory):
    """images.py:24-32: every *.png of `directory`, sorted, as float32 in [0,1]: [n, H, W(, C)]"""
    from PIL import Image
    out = []
    for path in sorted(glob.glob(os.path.join(directory, '*.png'))):
        a = np.asarray(Image.open(path))
        out.append(a.astype(np.float32) / (65535.0 if a.dtype == np.uint16 else 255.0))
    print("Loaded {} images from {}".format(len(out), directory))
    return np.asarray(out)


def load_train_data(directory):
    """images.py:240-253"""
    return load(os.path.abspath(os.path.join(directory, 'images/'))), load(os.path.abspath(os.path.join(directory, 'groundtruth/')))


def mirror_border(images, n):
    """images.py:269-281 (numpy 'symmetric' pad)"""
    pad = ((0, 0), (n, n), (n, n)) + (((0, 0),) if images.ndim == 4 else ())
    return np.pad(images, pad, "symmetric")


def extract_patches(images, patch_size, stride=None, predict_patch_size=None):
    """images.py:35-85 (host version for the training pool; the prediction path uses the fused device kernel)"""
    if not predict_patch_size:
        predict_patch_size = patch_size
    assert (patch_size - predict_patch_size) % 2 == 0 and predict_patch_size <= patch_size
    if not stride:
        stride = patch_size
    n, h, w = images.shape[:3]
    assert h == w, "Assume square images"
    assert (h - patch_size) % stride == 0, "Stride sliding should cover the whole image"
    starts = range(0, h - patch_size + 1, stride)
    out = np.zeros((n * len(starts) ** 2, patch_size, patch_size) + images.shape[3:])
    k = 0
    for i in range(n):
        for x in starts:          # x outer
            for y in starts:      # y inner
                out[k] = images[i, y:y + patch_size, x:x + patch_size]
                k += 1
    return out


def expand_and_rotate(imgs, angles, offset=0):
    """images.py:320-351: mirror-pad by ceil(h(sqrt2-1)/2) + ceil(offset/sqrt2), rotate every image by each angle with
    nearest-neighbour resampling (scipy.ndimage.rotate order=0, angle 0 skipped), centre-crop to h + 2*offset."""
    from scipy.ndimage import rotate
    has_channels = imgs.ndim == 4
    if not has_channels:
        imgs = imgs[..., None]
    b, h, w, c = imgs.shape
    assert h == w
    out_size = h + 2 * offset
    assert out_size % 2 == 0
    padding = int(np.ceil(h * (np.sqrt(2) - 1) / 2)) + int(np.ceil(offset / np.sqrt(2)))
    print("Applying rotations: {} degrees... ".format(", ".join(str(a) for a in angles)))
    padded = mirror_border(imgs, padding)
    res = np.zeros((b * len(angles), out_size, out_size, c))
    for i, angle in enumerate(angles):
        r = padded if angle == 0 else rotate(padded, angle=angle, axes=(1, 2), order=0)
        ctr, half = r.shape[1] // 2, out_size // 2
        res[i * b:(i + 1) * b] = r[:, ctr - half:ctr + half, ctr - half:ctr + half]
    return res if has_channels else res[..., 0]


def quantize_mask(masks, threshold, patch_size):
    """images.py:256-266: per patch_size block, label = mean(mask >= 0.5) > threshold"""
    out = masks.copy()
    n, size = masks.shape[0], masks.shape[1]
    for y in range(0, size, patch_size):
        for x in range(0, size, patch_size):
            lab = (masks[:, y:y + patch_size, x:x + patch_size, 0] >= 0.5).reshape(n, -1).mean(axis=1) > threshold
            out[:, y:y + patch_size, x:x + patch_size, 0] = lab[:, None, None]
    return out


def labels_for_patches(patches):
    """images.py:88-99"""
    return (patches.mean(axis=(1, 2)) > FOREGROUND_THRESHOLD).astype(np.int64)


def submission_rows(masks, patch_size=IMG_PATCH_SIZE):
    """body of images.save_submission_csv (images.py:206-237): '{img:03d}_{x}_{y},{label}' with x the outer index"""
    if masks.ndim == 4:
        masks = masks.squeeze(-1)
    n, h, w = masks.shape
    assert h == w, "images should be square"
    pps = h // patch_size
    labels = labels_for_patches(extract_patches(masks, patch_size)).reshape(n, pps, pps)
    return ["{:03d}_{}_{},{}".format(k + 1, patch_size * j, patch_size * i, labels[k, j, i])
            for k in range(n) for j in range(pps) for i in range(pps)]


def save_submission_csv(masks, path, patch_size=IMG_PATCH_SIZE):
    os.makedirs(path, exist_ok=True)
    filename = os.path.abspath(os.path.join(path, "submission.csv"))
    with open(filename, "w") as f:
        print("Saving predictions in {}".format(filename))
        f.write("id,prediction\n")
        for r in submission_rows(masks, patch_size):
            f.write(r + "\n")
    return filename
