"""-m gpu: parity at the sizes and on the data the BASELINE configurations name.
  * real aerial crops from the reference's own training set (tests/golden/real_crops.npz): train on 12, hold 6 out, and require the
    pixel-F1 of the HIP path within 1e-3 of the float32 oracle run with the same trained weights (the north_star tolerance);
  * config 3 (num_layers=6 root_size=64 patch_size=388 --dilated_layers) forward at full size, one patch, against the oracle;
  * config 4's per-GPU share (num_layers=6, four patches): a full-size training step -- properties only (the oracle would need
    hours): finite, repeatable bit for bit, loss falls on a repeated batch;
  * config 5 (604x604, stride 12, 6-way ensemble, 2166 tiles): shared-window masks == tile-by-tile masks, one tile against the
    oracle forward.
The network oracle stays parity-unpinned at the TensorFlow boundary (oracle/unet_oracle.py header); these tests pin the HIP path to
it, not to TensorFlow."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import tiler_oracle as T  # noqa: E402
from oracle import unet_oracle as U  # noqa: E402
from road_segmentation_unet_amd.model import ConvolutionalModel, Options, pixel_f1  # noqa: E402
from road_segmentation_unet_amd.unet import UNet  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def _predict_hip(net, xs, B):
    net.training = False
    out = []
    for i in range(0, xs.shape[0], B):
        xb = xs[i:i + B]
        net.x.zero_()
        net.x[:xb.shape[0]].copy_(torch.from_numpy(xb.astype(np.float32)))
        net.forward_device()
        out.append(net.prob[:xb.shape[0]].cpu().numpy().copy())
    return np.concatenate(out)


def test_real_crops_pixel_f1_within_1e3_of_the_float32_oracle():
    """Held-out real crops through a TRAINED network (tests/golden/real_crops_trained_params.npz: the L=3, root=32 net after 90 epochs
    on the 12 training crops, written by tests/golden/make_trained_params.py): pixel-F1 of the HIP path within 1e-3 of the float32
    oracle with the same weights. (Training inside the test made the F1 hinge on the summation order of the weight gradients: 0.38,
    0.15 and 0.0 after 240 steps for three builds that differ in nothing else -- see the next test for the training side.)"""
    z = np.load(os.path.join(HERE, "golden", "real_crops.npz"))
    x = z["x"].astype(np.float64) / 255.0
    y = (z["y"].astype(np.float64) / 255.0 >= 0.5) * 1.0
    P, S = int(z["P"]), int(z["S"])
    L, root, B = 3, 32, 4
    assert S == U.input_size_needed(P, L)
    xte, yte = x[12:], y[12:]
    pz = np.load(os.path.join(HERE, "golden", "real_crops_trained_params.npz"))
    params = {k.replace("__", "/"): pz[k] for k in pz.files}
    net = UNet(L, root, False, B, P, params=params, training=False)
    hip = _predict_hip(net, xte, B)
    ref = U.predict_probs(params, xte.astype(np.float32), L, root, False)                       # float32 oracle
    emu = U.predict_probs(params, xte.astype(np.float32), L, root, False, emulate_bf16=True)   # same rounding points as the HIP path
    f_hip, f_ref, f_emu = pixel_f1(hip, yte), pixel_f1(ref, yte), pixel_f1(emu, yte)
    print("pixel-F1 on 6 held-out real crops: hip %.5f  fp32 oracle %.5f  bf16-emulating oracle %.5f" % (f_hip, f_ref, f_emu))
    assert f_ref > 0.3, f_ref                      # the road class was actually learned on real data
    assert np.abs(hip - emu).max() <= 4e-3
    assert abs(f_hip - f_ref) <= 1e-3, (f_hip, f_ref)


def test_real_crops_training_then_inference_matches_the_rounding_emulating_oracle():
    """The training side on real data: 120 HIP steps from Glorot weights bring the loss down, and the network they produce predicts
    the held-out crops like the bf16-emulating oracle does with the same weights (wherever the chaotic trajectory has led)."""
    z = np.load(os.path.join(HERE, "golden", "real_crops.npz"))
    x = z["x"].astype(np.float64) / 255.0
    y = (z["y"].astype(np.float64) / 255.0 >= 0.5) * 1.0
    P, L, root, B = int(z["P"]), 3, 32, 4
    xtr, ytr, xte = x[:12], y[:12], x[12:]
    m = ConvolutionalModel(Options(num_layers=L, root_size=root, patch_size=P, batch_size=B, dropout=1.0, lr=0.02, seed=21, logdir=None))
    # the reference's loop drops the tail batch: 13 patches -> 3 steps of 4 per epoch (one crop doubled)
    xtr13, ytr13 = np.concatenate([xtr, xtr[:1]]), np.concatenate([ytr, ytr[:1]])
    first = m.train(xtr13, ytr13, None, None)["loss"]
    for _ in range(39):
        st = m.train(xtr13, ytr13, None, None)
    assert np.isfinite(st["loss"]) and st["loss"] < first, (first, st["loss"])
    params = {k: v for k, v in m.net.state_dict().items() if not k.endswith("/Momentum") and k != "global_step"}
    hip = _predict_hip(m.net, xte, B)
    emu = U.predict_probs(params, xte.astype(np.float32), L, root, False, emulate_bf16=True)
    assert np.abs(hip - emu).max() <= 4e-3


def test_config3_full_size_dilated_forward_matches_oracle():
    L, root, P = 6, 64, 388
    S = U.input_size_needed(P, L)
    assert S == 764
    rng = np.random.RandomState(31)
    x = rng.rand(1, S, S, 3).astype(np.float32)
    params = U.init_params(L, root, True, seed=32, bias_scale=0.02)
    net = UNet(L, root, True, 1, P, params=params, training=False)
    net.x.copy_(torch.from_numpy(x))
    net.forward_device()
    hip = net.prob.cpu().numpy().copy()
    emu = U.predict_probs(params, x, L, root, True, emulate_bf16=True)
    assert hip.shape == emu.shape == (1, P, P)
    assert np.isfinite(hip).all()
    assert np.abs(hip - emu).max() <= 4e-3, float(np.abs(hip - emu).max())


def test_config4_share_full_size_training_step_properties():
    L, root, P, B = 6, 64, 388, 4
    S = U.input_size_needed(P, L)
    g = torch.Generator().manual_seed(41)
    x = torch.rand((B, S, S, 3), generator=g)
    labels = (torch.rand((B, P, P), generator=g) < 0.2).to(torch.int64)
    finals, losses = [], []
    for rep in range(2):
        net = UNet(L, root, False, B, P, seed=42, training=True)
        net.x.copy_(x)
        net.labels.copy_(labels)
        ls = []
        for _ in range(3):
            net.forward_device()
            net.backward_device(1.0 / (B * P * P))
            ls.append(float(net.loss_sum.item()) / (B * P * P))
            assert torch.isfinite(net.flat_g[:net.n_live]).all()
            net.apply_momentum(0.01, 0.9)
        torch.cuda.synchronize()
        finals.append(net.flat_w.clone())
        losses.append(ls)
        del net
        torch.cuda.empty_cache()
    assert losses[0] == losses[1], losses                 # bit-repeatable, tile-shape tuning and two streams included
    assert torch.equal(finals[0], finals[1])
    assert all(np.isfinite(losses[0])) and losses[0][2] < losses[0][0], losses[0]
    assert abs(losses[0][0] - np.log(2.0)) < 0.3          # Glorot weights: the first loss sits near ln 2


def test_config5_full_size_sliding_window():
    L, root, P, H, stride = 6, 64, 388, 604, 12
    rng = np.random.RandomState(51)
    img = rng.rand(1, H, H, 3).astype(np.float32)
    params = U.init_params(L, root, True, seed=52, bias_scale=0.02)
    m = ConvolutionalModel(Options(num_layers=L, root_size=root, patch_size=P, stride=stride, dilated_layers=True, batch_size=4,
                                   ensemble_prediction=True, dropout=1.0, logdir=None), params=params)
    old = os.environ.get("RSU_PREDICT_SHARED")
    try:
        os.environ["RSU_PREDICT_SHARED"] = "1"
        shared = m.predict(img)
        os.environ["RSU_PREDICT_SHARED"] = "0"
        tilewise = m.predict(img)
    finally:
        if old is None:
            os.environ.pop("RSU_PREDICT_SHARED", None)
        else:
            os.environ["RSU_PREDICT_SHARED"] = old
    pps = (H - P) // stride + 1
    assert pps == 19 and 6 * pps * pps == 2166
    assert shared.shape == tilewise.shape == (1, H, H, 1)
    assert shared.min() >= 0.0 and shared.max() <= 1.0 and np.isfinite(shared).all()
    np.testing.assert_allclose(shared, tilewise, rtol=0, atol=5e-7)     # same tiles, fp32 association of the overlap sum only
    # one tile of the 2166 against the oracle forward: variant 3 (rot90) of the ensemble, tile (x index 7, y index 11)
    S = m.input_size
    ens = T.image_augmentation_ensemble(img)
    tiles = T.extract_patches(T.mirror_border(ens[3:4], (S - P) // 2), S, stride=stride, predict_patch_size=P)
    t = 7 * pps + 11
    emu = U.predict_probs(params, tiles[t:t + 1].astype(np.float32), L, root, True, emulate_bf16=True)
    m.net.training = False
    m.net.x.zero_()
    m.net.x[0].copy_(torch.from_numpy(tiles[t].astype(np.float32)))
    m.net.forward_device()
    got = m.net.prob[0].cpu().numpy()
    assert np.abs(got - emu[0]).max() <= 4e-3
