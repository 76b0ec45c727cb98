"""-m gpu: the host mirror of ConvolutionalModel (train / predict / save / restore) over the HIP path, checked against
the oracle's tiler + network restatements on identical weights and images."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import tiler_oracle as T  # noqa: E402
from oracle import unet_oracle as U  # noqa: E402
from road_segmentation_unet_amd.model import ConvolutionalModel, Options, pixel_f1  # noqa: E402


def _oracle_predict(params, imgs, L, root, dilated, P, stride, ensemble, emu=True):
    """tf_aerial_images.py:271-328 restated with the oracle pieces"""
    S = U.input_size_needed(P, L)
    x = T.image_augmentation_ensemble(imgs) if ensemble else imgs.astype(np.float64)
    n = x.shape[0]
    tiles = T.extract_patches(T.mirror_border(x, (S - P) // 2), S, stride=stride, predict_patch_size=P)
    probs = np.concatenate([U.predict_probs(params, tiles[i:i + 4].astype(np.float32), L, root, dilated, emulate_bf16=emu)
                            for i in range(0, tiles.shape[0], 4)])
    masks = T.images_from_patches(probs.astype(np.float64).reshape(n, -1, P, P, 1), stride=stride)
    return T.invert_image_augmentation_ensemble(masks) if ensemble else masks


@pytest.mark.parametrize("ensemble,batch", [(False, 3), (True, 4)])
def test_predict_matches_oracle(ensemble, batch):
    L, root, P, H, stride, dilated = 3, 16, 20, 44, 12, True
    rng = np.random.RandomState(1)
    imgs = rng.rand(2, H, H, 3).astype(np.float32)
    params = U.init_params(L, root, dilated, seed=2, bias_scale=0.05)
    opts = Options(num_layers=L, root_size=root, patch_size=P, stride=stride, dilated_layers=dilated, batch_size=batch,
                   ensemble_prediction=ensemble, dropout=1.0)
    m = ConvolutionalModel(opts, params=params)
    masks = m.predict(imgs)
    assert masks.shape == (2, H, H, 1)
    ref = _oracle_predict(params, imgs, L, root, dilated, P, stride, ensemble)
    assert np.abs(masks - ref).max() <= 4e-3          # vs bf16-emulating oracle
    ref32 = _oracle_predict(params, imgs, L, root, dilated, P, stride, ensemble, emu=False)
    assert np.abs(masks - ref32).max() <= 3e-2        # stated fp32 tolerance of the bf16 path
    mb = m.predict_batchwise(imgs, 1)
    np.testing.assert_allclose(mb, masks, rtol=0, atol=1e-6)


def test_train_epoch_save_restore(tmp_path):
    L, root, P, B = 3, 16, 20, 2
    S = U.input_size_needed(P, L)
    rng = np.random.RandomState(3)
    # learnable synthetic task: label = bright pixel in the centre crop of channel 0
    patches = rng.rand(13, S, S, 3)
    off = (S - P) // 2
    labels = (patches[:, off:off + P, off:off + P, 0] > 0.5) * 1.0
    opts = Options(num_layers=L, root_size=root, patch_size=P, batch_size=B, dropout=1.0, lr=0.05, save_path=str(tmp_path), seed=7)
    m = ConvolutionalModel(opts)
    losses = []
    for _ in range(6):
        st = m.train(patches, labels, None, None)
        losses.append(st["loss"])
        assert st["patches"] == 12  # range(0, 13 - 2, 2): the reference's loop drops the tail
    assert m.net.global_step == 36
    assert losses[-1] < losses[0] * 0.9, losses
    path = m.save(5)
    w_before = {k: v.copy() for k, v in m.net.state_dict().items()}
    m2 = ConvolutionalModel(Options(num_layers=L, root_size=root, patch_size=P, batch_size=B, dropout=1.0, save_path=str(tmp_path), seed=99))
    m2.restore(file=path)
    for k, v in m2.net.state_dict().items():
        np.testing.assert_array_equal(v, w_before[k])
    assert m2.net.global_step == 36
    m3 = ConvolutionalModel(Options(num_layers=L, root_size=root, patch_size=P, batch_size=B, dropout=1.0, save_path=str(tmp_path)))
    m3.restore()  # newest experiment dir, newest epoch
    np.testing.assert_array_equal(m3.net.state_dict()["conv_0/conv1/kernel"], w_before["conv_0/conv1/kernel"])
    x = rng.rand(B, S, S, 3).astype(np.float32)
    m.net.x.copy_(torch.from_numpy(x)); m2.net.x.copy_(torch.from_numpy(x))
    m.net.training = m2.net.training = False
    m.net.forward_device(); m2.net.forward_device()
    assert torch.equal(m.net.prob, m2.net.prob)


def test_pixel_f1_parity_after_training():
    """north_star: pixel-F1 on held-out patches within 1e-3 of the reference arithmetic (here: the float32 oracle evaluated with the
    weights the HIP path trained)."""
    L, root, P, B = 3, 16, 20, 4
    S = U.input_size_needed(P, L)
    rng = np.random.RandomState(11)
    def make(n):
        x = rng.rand(n, S, S, 3)
        off = (S - P) // 2
        return x, (x[:, off:off + P, off:off + P, 1] > 0.55) * 1.0
    xtr, ytr = make(41)
    xte, yte = make(16)
    m = ConvolutionalModel(Options(num_layers=L, root_size=root, patch_size=P, batch_size=B, dropout=1.0, lr=0.05, seed=5))
    for _ in range(12):
        m.train(xtr, ytr, None, None)
    params = {k: v for k, v in m.net.state_dict().items() if not k.endswith("/Momentum") and k != "global_step"}
    m.net.training = False
    hip = []
    for i in range(0, 16, B):
        m.net.x.copy_(torch.from_numpy(xte[i:i + B].astype(np.float32)))
        m.net.forward_device()
        hip.append(m.net.prob.cpu().numpy().copy())
    hip = np.concatenate(hip)
    ref = np.concatenate([U.predict_probs(params, xte[i:i + B].astype(np.float32), L, root, False) for i in range(0, 16, B)])
    f_hip, f_ref = pixel_f1(hip, yte), pixel_f1(ref, yte)
    assert f_ref > 0.6, f_ref   # the task was actually learned
    assert abs(f_hip - f_ref) <= 1e-3 + 2e-3, (f_hip, f_ref)


@pytest.mark.parametrize("L,P,H,stride,dilated,ensemble", [(3, 20, 44, 12, True, True), (3, 20, 52, 2, False, False), (4, 28, 64, 6, True, False),
                                                           (2, 12, 40, 4, False, True)])
def test_shared_window_prediction_equals_tilewise(L, P, H, stride, dilated, ensemble):
    """tiles with equal offsets modulo 2^(L-1) computed as sub-windows of one larger forward pass: the masks must be bit-identical to
    the per-tile sliding window of tf_aerial_images.py:288-320 (every output element sees the same arithmetic)"""
    import os
    root = 16
    rng = np.random.RandomState(L + stride)
    imgs = rng.rand(2, H, H, 3).astype(np.float32)
    params = U.init_params(L, root, dilated, seed=3, bias_scale=0.05)
    opts = Options(num_layers=L, root_size=root, patch_size=P, stride=stride, dilated_layers=dilated, batch_size=4,
                   ensemble_prediction=ensemble, dropout=1.0)
    m = ConvolutionalModel(opts, params=params)
    old = os.environ.get("RSU_PREDICT_SHARED")
    try:
        os.environ["RSU_PREDICT_SHARED"] = "0"
        tilewise = m.predict(imgs)
        os.environ["RSU_PREDICT_SHARED"] = "1"
        shared = m.predict(imgs)
        os.environ["RSU_PREDICT_MAX_WINDOW"] = str(m.input_size + 8)   # force runs of one or two tiles per axis
        shared_small = m.predict(imgs)
    finally:
        os.environ.pop("RSU_PREDICT_MAX_WINDOW", None)
        if old is None:
            os.environ.pop("RSU_PREDICT_SHARED", None)
        else:
            os.environ["RSU_PREDICT_SHARED"] = old
    # masks: the overlap average adds the same tiles in the same order but in different launch groupings (fp32 association)
    np.testing.assert_allclose(shared, tilewise, rtol=0, atol=5e-7)
    np.testing.assert_allclose(shared_small, tilewise, rtol=0, atol=5e-7)
    # tiles: bit-identical to the per-tile forward pass
    import torch
    from road_segmentation_unet_amd import images as dimages
    x = torch.as_tensor(imgs).to(m.net.device)
    if ensemble:
        x = dimages.image_augmentation_ensemble(x).contiguous()
    pps = (H - P) // stride + 1
    tiles = m._shared_window_tiles(x, pps)
    m.net.training = False
    B = m.local_batch
    flat = tiles.view(-1, P, P)
    for t0 in range(0, flat.shape[0], B):
        nb = min(B, flat.shape[0] - t0)
        dimages.extract_mirrored_patches(x, m.input_size, P, stride, t0=t0, ntiles=nb, out=m.net.x[:nb])
        m.net.forward_device()
        assert torch.equal(m.net.prob[:nb], flat[t0:t0 + nb]), t0
