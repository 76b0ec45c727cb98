"""Host-side data formats either side of the hot path (SURVEY.md section 8(f) "next" rows 2 and 3): PNG loading, the
offline rotation augmentation, mask quantisation and the Kaggle submission CSV. Plain numpy/scipy/PIL, off the timed path;
each function cites the reference lines it mirrors (/root/reference/src/images.py)."""
import glob
import os

import numpy as np

FOREGROUND_THRESHOLD = .25  # src/constants.py:1
IMG_PATCH_SIZE = 16         # src/constants.py:2
PIXEL_DEPTH = 255           # src/constants.py:5


def img_float_to_uint8(img):
    """images.py:19-21"""
    return (np.asarray(img) * PIXEL_DEPTH).round().astype(np.uint8)


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


def overlays(imgs, masks, fade=0.95):
    """Road masks painted in red over the aerial images (the reference's images.overlays, images.py:102-128): per image one
    PIL alpha-composite of an RGBA layer (red, alpha = 255 * mask * fade, truncated to uint8) onto the RGB image made opaque.
    imgs [n, H, W, 3] float in [0, 1], masks [n, H, W(, 1)] -> uint8 [n, H, W, 4]."""
    from PIL import Image
    pictures = img_float_to_uint8(np.asarray(imgs))
    if pictures.ndim != 4 or pictures.shape[-1] != 3:
        raise AssertionError('Predict image should be colored')
    n, height, width = pictures.shape[:3]
    alpha = (img_float_to_uint8(np.asarray(masks).reshape(n, height, width)) * fade).astype(np.uint8)

    def painted(k):
        layer = np.zeros((height, width, 4), dtype=np.uint8)
        layer[..., 0] = 255
        layer[..., 3] = alpha[k]
        base = Image.fromarray(pictures[k]).convert("RGBA")
        return np.asarray(Image.alpha_composite(base, Image.fromarray(layer)))

    return np.stack([painted(k) for k in range(n)]) if n else np.zeros((0, height, width, 4), dtype=np.uint8)


def overlap_pred_true(pred, true):
    """images.py:282-293: prediction in the red, ground truth in the green channel"""
    pred, true = np.asarray(pred), np.asarray(true)
    num_images, im_height, im_width = pred.shape
    out = np.zeros((num_images, im_height, im_width, 3), dtype=np.uint8)
    out[:, :, :, 0] = img_float_to_uint8(pred)
    out[:, :, :, 1] = img_float_to_uint8(true)
    return out


def overlapp_error(pred, true):
    """images.py:296-309: white where prediction and ground truth agree"""
    pred, true = np.asarray(pred), np.asarray(true)
    num_images, im_height, im_width = pred.shape
    agree = np.logical_not(np.logical_xor(img_float_to_uint8(true).astype(bool), img_float_to_uint8(pred).astype(bool)))
    err = img_float_to_uint8(agree * 1)
    return np.repeat(err[..., None], 3, axis=-1)


def save_all(images, directory, format_="images_{:03d}.png", greyscale=False):
    """images.py:185-205 (matplotlib.image.imsave: 2-D arrays go through the colour map, normalised to their own range)"""
    import matplotlib as mpl
    import matplotlib.image as mpimg
    images = np.asarray(images)
    os.makedirs(directory, exist_ok=True)
    if images.ndim == 4 and images.shape[-1] == 1:
        images = images.squeeze(-1)
    cmap = "gray" if greyscale else mpl.rcParams.get("image.cmap")
    for n in range(images.shape[0]):
        mpimg.imsave(os.path.join(directory, format_.format(n + 1)), images[n], cmap=cmap)


def img_to_label_patches(img, patch_size=IMG_PATCH_SIZE):
    """summary.py:134-139 on the host, quirk included: the [n] label vector resized in place to [n, ps, ps] (zero filled)"""
    lab = labels_for_patches(extract_patches(np.asarray(img), patch_size))
    out = np.zeros(lab.shape[0] * patch_size * patch_size, dtype=lab.dtype)
    out[:lab.shape[0]] = lab
    return out.reshape(lab.shape[0], patch_size, patch_size)
