"""-m gpu: the rows next to the hot path that round 2 added on the device: mask quantisation and patch
labels against reference-generated goldens, the streaming metrics of summary.py, the training-patch pool / uploader / D4
augmentation, the TF-array checkpoint importer and the --eval_train branch of the command line."""
import ctypes
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import unet_oracle as U  # noqa: E402
from road_segmentation_unet_amd import hostio  # noqa: E402
from road_segmentation_unet_amd._lib import call  # noqa: E402
from road_segmentation_unet_amd.model import ConvolutionalModel, Options  # noqa: E402
from road_segmentation_unet_amd.pool import DevicePatchPool, PatchPool, d4_apply  # noqa: E402
from road_segmentation_unet_amd.summary import StreamingMetrics, Summary  # noqa: E402
from tests import hiputil as hu  # noqa: E402


def test_quantize_mask_and_patch_labels_match_reference_goldens(golden):
    for key_in, key_out in (("g5_mask_in", "g5_quant"), ("g5_mask2_in", "g5_quant2")):
        m = golden[key_in]
        t = hu.dev_f32(m[..., 0])
        out = torch.empty_like(t)
        call("rsu_quantize_mask", hu.ptr(t), hu.ptr(out), t.shape[0], t.shape[1], 16, 0.25, hu.stream())
        np.testing.assert_array_equal(hu.host(out), golden[key_out][..., 0].astype(np.float32))
        call("rsu_quantize_mask", hu.ptr(t), hu.ptr(t), t.shape[0], t.shape[1], 16, 0.25, hu.stream())  # in place
        np.testing.assert_array_equal(hu.host(t), golden[key_out][..., 0].astype(np.float32))
    # edge blocks: 40-pixel masks with 16-pixel blocks average over the pixels they hold, like numpy slicing
    rng = np.random.RandomState(0)
    m = rng.rand(2, 40, 40, 1)
    t = hu.dev_f32(m[..., 0])
    call("rsu_quantize_mask", hu.ptr(t), hu.ptr(t), 2, 40, 16, 0.25, hu.stream())
    np.testing.assert_array_equal(hu.host(t), hostio.quantize_mask(m, 0.25, 16)[..., 0].astype(np.float32))
    # labels_for_patches(extract_patches(mask, 16)) in the reference's patch order
    lab_in = golden["g10_lab_in"]
    t = hu.dev_f32(lab_in)
    nb = lab_in.shape[1] // 16
    lab = torch.empty((lab_in.shape[0], nb, nb), dtype=torch.int64, device=hu.DEV)
    call("rsu_labels_for_patches", hu.ptr(t), hu.ptr(lab), lab_in.shape[0], lab_in.shape[1], 16, 0.25, hu.stream())
    n = lab.numel()
    np.testing.assert_array_equal(lab.cpu().numpy().reshape(-1), golden["g10_label_patches"].reshape(-1)[:n])


def test_streaming_metrics_and_summary(tmp_path, golden):
    rng = np.random.RandomState(1)
    sm = StreamingMetrics(hu.DEV)
    tp = fp = fn = tn = 0
    for _ in range(3):  # the counters run on over calls, like tf.metrics' local variables
        p, t = (rng.rand(5000) > 0.6).astype(np.int64), (rng.rand(5000) > 0.5).astype(np.int64)
        acc, rec, prec, f1 = sm.update(torch.from_numpy(p).to(hu.DEV), torch.from_numpy(t).to(hu.DEV))
        tp += int(((p == 1) & (t == 1)).sum()); fp += int(((p == 1) & (t == 0)).sum())
        fn += int(((p == 0) & (t == 1)).sum()); tn += int(((p == 0) & (t == 0)).sum())
        r, q = tp / (tp + fn), tp / (tp + fp)
        assert (acc, rec, prec) == ((tp + tn) / (tp + fp + fn + tn), r, q)
        assert f1 == 2 / (1 / r + 1 / q)   # summary.py:145
    sm.reset()
    z = torch.zeros(10, dtype=torch.int64, device=hu.DEV)
    assert sm.update(z, z) == (1.0, 0.0, 0.0, 0.0)   # tf.metrics give 0 for 0/0; 1/0 = inf makes the F1 0
    opts = Options(num_eval_images=2, logdir=str(tmp_path))
    s = Summary(opts, None, str(tmp_path / "run"), device=hu.DEV)
    s.initialize_train_summary()
    s.initialize_eval_summary()
    masks = golden["g10_lab_in"]
    truth = (rng.rand(*masks.shape) > 0.7) * 1.0
    acc, rec, prec, f1 = s.add_to_training_summary(masks, truth, 7)
    # the same numbers from the host mirror of img_to_label_patches (zero-padded to [n,16,16], which only the accuracy sees)
    pl, tl = hostio.img_to_label_patches(masks).reshape(-1), hostio.img_to_label_patches(truth).reshape(-1)
    tp_, fp_, fn_ = int(((pl == 1) & (tl == 1)).sum()), int(((pl == 1) & (tl == 0)).sum()), int(((pl == 0) & (tl == 1)).sum())
    assert acc == float((pl == tl).mean())
    assert rec == (tp_ / (tp_ + fn_) if tp_ + fn_ else 0.0) and prec == (tp_ / (tp_ + fp_) if tp_ + fp_ else 0.0)
    s.add({"loss": 0.5, "learning_rate": 0.01}, global_step=7)
    s.add_to_pixel_missclassification_summary(30.0, 8, 7)
    s.add_to_overlap_summary(truth[:2], (masks[:2] > 0.3) * 1, 7)
    s.flush()
    ev = [json.loads(l) for l in open(tmp_path / "run" / "events.jsonl")]
    tags = {e["tag"] for e in ev}
    assert {"train accuracy", "train recall", "train precision", "train f1_score", "loss", "learning_rate", "misclassification_rate"} <= tags
    assert [e["value"] for e in ev if e["tag"] == "misclassification_rate"] == [30.0 / 8]
    assert any(f.startswith("step_0000007_groundtruth_vs_prediction") for f in os.listdir(tmp_path / "run"))


def _pool_inputs(rng, n=3, E=44, S=28, P=12):
    off = (S - P) // 2
    ext = rng.rand(n, E, E, 3)
    lab = (rng.rand(n, E - 2 * off, E - 2 * off) > 0.5) * 1.0
    return ext, lab, S, P


def test_patch_pools_hold_the_reference_patches():
    rng = np.random.RandomState(2)
    ext, lab, S, P = _pool_inputs(rng)
    stride = 4
    ref_x = hostio.extract_patches(ext, patch_size=S, predict_patch_size=P, stride=stride)
    ref_y = hostio.extract_patches(lab, patch_size=P, stride=stride)
    idx = rng.permutation(ref_x.shape[0])[:9]
    hp = PatchPool(ext, lab, S, P, stride)
    assert hp.shape == ref_x.shape and len(hp) == ref_x.shape[0]
    x, y = hp.gather(idx)
    np.testing.assert_array_equal(x, ref_x[idx].astype(np.float32))
    np.testing.assert_array_equal(y, ref_y[idx].astype(np.float32))
    dp = DevicePatchPool(ext, lab, S, P, stride, device=hu.DEV)
    xo = torch.empty((9, S, S, 3), dtype=torch.float32, device=hu.DEV)
    yo = torch.empty((9, P, P), dtype=torch.int64, device=hu.DEV)
    assert dp.load_batch(idx, xo, yo) is None
    np.testing.assert_array_equal(xo.cpu().numpy(), ref_x[idx].astype(np.float32))
    np.testing.assert_array_equal(yo.cpu().numpy(), ref_y[idx].astype(np.int64))


def test_d4_augmentation_moves_image_and_label_together():
    rng = np.random.RandomState(3)
    ext, _, S, P = _pool_inputs(rng)
    off = (S - P) // 2
    lab = (ext[:, off:-off, off:-off, 0] > 0.5) * 1.0    # the label IS a function of the centre crop: alignment is checkable
    dp = DevicePatchPool(ext, lab, S, P, 4, device=hu.DEV, augment=True, seed=11)
    idx = np.arange(0, len(dp), 3)[:16]
    xo = torch.empty((len(idx), S, S, 3), dtype=torch.float32, device=hu.DEV)
    yo = torch.empty((len(idx), P, P), dtype=torch.int64, device=hu.DEV)
    ops = dp.load_batch(idx, xo, yo)
    assert len(set(ops)) > 4   # a spread of D4 elements was drawn
    x, y = xo.cpu().numpy(), yo.cpu().numpy()
    np.testing.assert_array_equal((x[:, off:-off, off:-off, 0] > 0.5) * 1, y)
    plain = PatchPool(ext, lab, S, P, 4).gather(idx)[0]
    for j, (ud, lr, tr, k) in enumerate(ops):   # tf.image.flip_up_down / flip_left_right / transpose_image / rot90 in numpy
        a = plain[j]
        a = a[::-1] if ud else a
        a = a[:, ::-1] if lr else a
        a = a.transpose(1, 0, 2) if tr else a
        a = np.rot90(a, k, (0, 1))
        np.testing.assert_array_equal(x[j], a)
    assert torch.equal(d4_apply(xo[0], (False, False, False, 0)), xo[0])


def test_three_input_paths_train_to_the_same_weights():
    """the reference's arrays through the pinned double-buffered uploader, the host index pool and the device pool feed the
    same batches: identical weights after an epoch"""
    L, root, P, B, stride = 2, 16, 12, 3, 8
    S = U.input_size_needed(P, L)
    rng = np.random.RandomState(5)
    off = (S - P) // 2
    ext = rng.rand(2, S + 2 * stride, S + 2 * stride, 3)
    lab = (ext[:, off:-off, off:-off, 1] > 0.5) * 1.0
    px = hostio.extract_patches(ext, patch_size=S, predict_patch_size=P, stride=stride)
    py = hostio.extract_patches(lab, patch_size=P, stride=stride)
    finals = []
    for kind in ("arrays", "host_pool", "device_pool"):
        m = ConvolutionalModel(Options(num_layers=L, root_size=root, patch_size=P, batch_size=B, dropout=0.8, lr=0.05, seed=9, logdir=None))
        np.random.seed(123)   # the shuffle of train() (np.random, like the reference)
        if kind == "arrays":
            st = m.train(px, py, None, None)
        elif kind == "host_pool":
            st = m.train(PatchPool(ext, lab, S, P, stride), None, None, None)
        else:
            st = m.train(DevicePatchPool(ext, lab, S, P, stride, device=m.net.device), None, None, None)
        assert st["patches"] == (px.shape[0] - 1) // B * B if px.shape[0] % B else st["patches"] > 0
        finals.append(m.net.flat_w.cpu().numpy().copy())
    np.testing.assert_array_equal(finals[0], finals[1])
    np.testing.assert_array_equal(finals[0], finals[2])


def test_restore_from_tf_arrays_is_a_pure_rename():
    L, root, P = 3, 16, 20
    a = ConvolutionalModel(Options(num_layers=L, root_size=root, patch_size=P, batch_size=2, dilated_layers=True, seed=1, logdir=None))
    sd = a.net.state_dict()
    exported = {("unet/" + k if not k.startswith("global") else k) + (":0" if i % 2 else ""): v for i, (k, v) in enumerate(sd.items())}
    exported["global_step"] = np.int64(1234)
    b = ConvolutionalModel(Options(num_layers=L, root_size=root, patch_size=P, batch_size=2, dilated_layers=True, seed=77, logdir=None))
    b.restore_from_tf_arrays(exported)
    for k, v in b.net.state_dict().items():
        if k != "global_step":
            np.testing.assert_array_equal(v, sd[k])
    assert b.net.global_step == 1234
    del exported[[k for k in exported if "weight_output/kernel" in k and "Momentum" not in k][0]]
    with pytest.raises(KeyError):
        b.restore_from_tf_arrays(exported)


def test_cli_eval_train_branch_and_overlays(tmp_path):
    from PIL import Image
    from road_segmentation_unet_amd.cli import main
    rng = np.random.RandomState(6)
    tr, ev = tmp_path / "train", tmp_path / "eval"
    (tr / "images").mkdir(parents=True)
    (tr / "groundtruth").mkdir(parents=True)
    ev.mkdir()
    H = 32
    for i in range(2):
        img = (rng.rand(H, H, 3) * 255).astype(np.uint8)
        Image.fromarray(img).save(tr / "images" / ("satImage_%03d.png" % i))
        Image.fromarray(((img[..., 0] > 127) * 255).astype(np.uint8)).save(tr / "groundtruth" / ("satImage_%03d.png" % i))
        Image.fromarray(img).save(ev / ("test_%d.png" % i))
    common = ["--num_layers=2", "--root_size=16", "--patch_size=16", "--stride=16", "--batch_size=2", "--lr=0.05", "--seed=5",
              "--train_data_dir=%s" % tr, "--save_path=%s" % (tmp_path / "runs"), "--logdir=%s" % (tmp_path / "log"), "--pred_batch_size=2"]
    out_dir = tmp_path / "evalout"
    assert main(common + ["--num_epoch=1", "--rotation_angles=0", "--eval_train", "--eval_data_dir=%s" % out_dir, "--d4_augmentation"]) == 0
    names = sorted(os.listdir(out_dir))
    for stem in ("eval_binary_pred_", "eval_probability_pred_", "eval_overlays_pred_", "eval_confusion_", "eval_orror_"):
        assert [n for n in names if n.startswith(stem)] == [stem + "001.png", stem + "002.png"], names
    logs = os.listdir(tmp_path / "log")
    assert len(logs) == 1 and "events.jsonl" in os.listdir(tmp_path / "log" / logs[0])
    # prediction branch: submission CSV + overlays + the model used
    assert main(common + ["--num_epoch=0", "--eval_data_dir=%s" % ev]) == 0
    runs = sorted(os.listdir(tmp_path / "runs"))
    sub = [r for r in runs if os.path.isdir(tmp_path / "runs" / r) and "submission.csv" in os.listdir(tmp_path / "runs" / r)]
    assert len(sub) == 1
    files = os.listdir(tmp_path / "runs" / sub[0])
    assert "images_001.png" in files and "images_002.png" in files
    assert any(r.endswith("-model.chkpt.npz") for r in runs), runs
