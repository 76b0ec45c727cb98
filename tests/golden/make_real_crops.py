"""Generates tests/golden/real_crops.npz from the reference's shipped training set (/root/reference/data/training, read-only): a few
REAL aerial crops with their ground truth, as data (SURVEY.md section 2 #12 allows committing crops). Build container only.

    python tests/golden/make_real_crops.py

Crops are S x S input windows (S = input_size_needed(P, 3) = P + 40) cut from the mirror-extended images exactly as the training
path does (expand by (S - P) / 2 with symmetric padding, tf_aerial_images.py:401-404), with the P x P ground truth under the centre.
uint8, as stored in the PNGs."""
import os

import numpy as np
from PIL import Image

REF = "/root/reference/data/training"
P, S = 92, 132
OFF = (S - P) // 2
rng = np.random.RandomState(2017)
names = sorted(os.listdir(os.path.join(REF, "images")))
pick = rng.choice(len(names), 18, replace=False)
xs, ys, src = [], [], []
for k in pick:
    img = np.asarray(Image.open(os.path.join(REF, "images", names[k])))
    gt = np.asarray(Image.open(os.path.join(REF, "groundtruth", names[k])))
    ext = np.pad(img, ((OFF, OFF), (OFF, OFF), (0, 0)), "symmetric")
    # a window with both classes present where possible
    for _ in range(50):
        y0, x0 = rng.randint(0, 400 - P + 1, 2)
        frac = (gt[y0:y0 + P, x0:x0 + P] >= 128).mean()
        if 0.08 < frac < 0.6:
            break
    xs.append(ext[y0:y0 + S, x0:x0 + S])
    ys.append(gt[y0:y0 + P, x0:x0 + P])
    src.append("%s@%d,%d" % (names[k], y0, x0))
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "real_crops.npz")
np.savez_compressed(out, x=np.stack(xs), y=np.stack(ys), src=np.array(src), P=np.int64(P), S=np.int64(S))
print("wrote", out, os.path.getsize(out), "bytes;", "road fraction %.3f" % (np.stack(ys) >= 128).mean())
