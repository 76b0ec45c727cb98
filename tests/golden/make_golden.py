"""Generates tests/golden/tiler_golden.npz by IMPORTING the reference (read-only at /root/reference).

Run in the build container only (the reference does not exist on the GPU box):
    python tests/golden/make_golden.py
Inputs are seeded; the .npz holds inputs and the reference's outputs (data only, no reference source).
Covers SURVEY.md section 8(c) G1-G6, G8 and a G9 real-data block from submissions/.
"""
import os
import sys
import types

import numpy as np

REF = "/root/reference"
sys.path.insert(0, os.path.join(REF, "src"))
sys.modules.setdefault("tensorflow", types.ModuleType("tensorflow"))  # unet.py only needs tf inside forward()
import images  # noqa: E402  (reference module)
import unet  # noqa: E402    (reference module; only input_size_needed is callable without TF)

out = {}
rng = np.random.RandomState(2017)

# G1 mirror_border 3-D and 4-D
a4 = rng.rand(2, 7, 7, 3).astype(np.float32)
a3 = rng.rand(2, 6, 6).astype(np.float32)
out["g1_in4"], out["g1_out4_n3"] = a4, images.mirror_border(a4, 3)
out["g1_in3"], out["g1_out3_n2"] = a3, images.mirror_border(a3, 2)
out["g1_out4_n7"] = images.mirror_border(a4, 7)

# G2 extract_patches: ordering probe + strided + predict_patch_size
probe = np.arange(36, dtype=np.float64).reshape(1, 6, 6)
out["g2_probe_in"], out["g2_probe_out"] = probe, images.extract_patches(probe, 2, stride=2)
b4 = rng.rand(2, 20, 20, 3).astype(np.float32)
out["g2_in4"] = b4
out["g2_out4_p8_s4"] = images.extract_patches(b4, 8, stride=4)
out["g2_out4_p12_s4_pp4"] = images.extract_patches(b4, 12, stride=4, predict_patch_size=4)
b3 = rng.rand(3, 12, 12).astype(np.float32)
out["g2_in3"], out["g2_out3_p4"] = b3, images.extract_patches(b3, 4)
out["g2_out3_p6_s3"] = images.extract_patches(b3, 6, stride=3)

# G3 images_from_patches: exact round trip and non-trivial overlap
pt = images.extract_patches(b4, 8, stride=4).reshape(2, -1, 8, 8, 3)
out["g3_roundtrip"] = images.images_from_patches(pt, stride=4)
pr = rng.rand(2, 9, 6, 6, 1)
out["g3_in"], out["g3_out_s3"] = pr, images.images_from_patches(pr, stride=3)
out["g3_out_nostride"] = images.images_from_patches(pr)

# G4 ensemble + inverse
e = rng.rand(2, 5, 5, 3).astype(np.float32)
out["g4_in"], out["g4_aug"] = e, images.image_augmentation_ensemble(e)
m = rng.rand(12, 5, 5, 1)
out["g4_masks"] = m.copy()
out["g4_inv"] = images.invert_image_augmentation_ensemble(m.copy())
out["g4_roundtrip"] = images.invert_image_augmentation_ensemble(images.image_augmentation_ensemble(e))

# G5 quantize_mask, labels_for_patches, csv rows
qm = rng.rand(2, 32, 32, 1)
out["g5_mask_in"] = qm
out["g5_quant"] = images.quantize_mask(qm, threshold=0.25, patch_size=16)
qm2 = (rng.rand(2, 48, 48, 1) > 0.7) * 1.0
out["g5_mask2_in"], out["g5_quant2"] = qm2, images.quantize_mask(qm2, threshold=0.25, patch_size=16)
lp = rng.rand(10, 16, 16) * 0.5
out["g5_lab_in"], out["g5_lab_out"] = lp, images.labels_for_patches(lp)
import tempfile  # noqa: E402
with tempfile.TemporaryDirectory() as d:
    images.save_submission_csv(out["g5_quant2"], d, 16)
    out["g5_csv"] = np.array(open(os.path.join(d, "submission.csv")).read())

# G6 input_size_needed
tab = []
for L in range(1, 7):
    for P in range(4, 400, 4):
        try:
            tab.append((L, P, unet.input_size_needed(P, L)))
        except AssertionError:
            tab.append((L, P, -1))
out["g6_table"] = np.array(tab, dtype=np.int64)
try:
    unet.input_size_needed(128, 5)
    out["g6_assert_msg"] = np.array("")
except AssertionError as ex:
    out["g6_assert_msg"] = np.array(str(ex))

# G8 predictions_to_patches (the reference's own value test, src/test_images.py:137-144)
pp = np.array([1, 0, 1, 1, 0], dtype=np.float64)
out["g8_in"], out["g8_out"] = pp, np.ascontiguousarray(images.predictions_to_patches(pp, 4))

# G9 real-data block: first image of a shipped submission; CSV rows + the per-block labels
sub = os.path.join(REF, "submissions")
sd = sorted(d for d in os.listdir(sub) if os.path.isdir(os.path.join(sub, d)))[-1]
rows = open(os.path.join(sub, sd, "submission.csv")).read().splitlines()
rows1 = [r for r in rows[1:] if r.startswith("001_")]
out["g9_dir"] = np.array(sd)
out["g9_rows_img1"] = np.array("\n".join(rows1))
lab = np.zeros((38, 38), np.int64)
for r in rows1:
    ident, v = r.split(",")
    _, a, b = ident.split("_")
    lab[int(a) // 16, int(b) // 16] = int(v)
out["g9_labels_img1"] = lab

# G7 expand_and_rotate (a newer scipy here; the reference pinned scipy 1.0.0 -- order=0 resampling may differ at exact ties)
import scipy  # noqa: E402
g7 = rng.rand(2, 40, 40, 3).astype(np.float32)
out["g7_in"] = g7
out["g7_out_off6"] = images.expand_and_rotate(g7, [0, 15, 45], 6)
g7m = rng.rand(2, 40, 40).astype(np.float32)
out["g7_in3"], out["g7_out3_off0"] = g7m, images.expand_and_rotate(g7m, [0, 30], 0)
out["g7_scipy"] = np.array(scipy.__version__)

# G10 post-processing images + the metrics' label patches (src/images.py:102-128,282-309; src/summary.py:134-139 with a stub tensorflow)
g10i = rng.rand(2, 32, 32, 3)
g10m = rng.rand(2, 32, 32, 1)
out["g10_imgs"], out["g10_masks"] = g10i, g10m
out["g10_overlays_f095"] = images.overlays(g10i, g10m)
out["g10_overlays_f04"] = images.overlays(g10i, g10m, fade=0.4)
g10p = (rng.rand(2, 32, 32) > 0.6) * 1
g10t = (rng.rand(2, 32, 32) > 0.5) * 1.0
out["g10_pred"], out["g10_true"] = g10p, g10t
out["g10_overlap"] = images.overlap_pred_true(g10p, g10t)
out["g10_error"] = images.overlapp_error(g10p, g10t)
out["g10_u8"] = images.img_float_to_uint8(g10m)
import summary  # noqa: E402  (reference module; tensorflow is the stub registered above)
g10l = rng.rand(3, 48, 48) * 0.6
out["g10_lab_in"] = g10l
out["g10_label_patches"] = summary.Summary.img_to_label_patches(None, g10l.copy())

# F: the 30 command-line flags (name, type, default) parsed from the reference's DEFINE_* calls -- data, not source
import json  # noqa: E402
import re  # noqa: E402
flags = []
srcdir = os.path.join(REF, "src")
os.chdir(srcdir)
class _F:  # minimal recorder standing in for tf.app.flags
    def __getattr__(self, name):
        if name.startswith("DEFINE_"):
            kind = name[len("DEFINE_"):]
            return lambda n, d, h: flags.append([n, kind, d])
        raise AttributeError(name)
tfm = sys.modules["tensorflow"]
tfm.app = types.SimpleNamespace(flags=_F(), run=lambda: None)
tfm.app.flags.FLAGS = types.SimpleNamespace()
sys.modules["summary"] = types.ModuleType("summary"); sys.modules["summary"].Summary = object
try:
    import tf_aerial_images  # noqa: E402,F401  (module-level DEFINE_* calls run against the recorder)
except Exception as ex:  # anything after the flag definitions is irrelevant here
    print("note: import stopped after flags:", type(ex).__name__)
here = os.path.abspath("/root/reference")
for f in flags:
    if isinstance(f[2], str) and f[2].startswith(here):
        f[2] = "<abs>" + f[2][len(here):]
out["flags_json"] = np.array(json.dumps(flags))
print("flags recorded:", len(flags))

np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "tiler_golden.npz"), **out)
print("wrote", len(out), "arrays")
