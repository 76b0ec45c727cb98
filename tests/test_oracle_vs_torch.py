"""Independent second opinion for oracle/unet_oracle.c: stock PyTorch-CPU float64 ops
(F.conv2d / conv_transpose2d / max_pool2d / cross_entropy + autograd). Not the reference itself
(TensorFlow 1.4 is not installable here) -- see the oracle header: parity unpinned at the TF boundary."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import unet_oracle as U

TOL = dict(rtol=2e-5, atol=2e-6)


def t64(a):
    return torch.from_numpy(np.asarray(a, dtype=np.float64))


def nhwc_to_nchw(a):
    return t64(a).permute(0, 3, 1, 2).contiguous()


@pytest.mark.parametrize("dil", [1, 2])
@pytest.mark.parametrize("cin,cout", [(3, 8), (8, 5)])
def test_conv2d_fwd_bwd(dil, cin, cout):
    rng = np.random.RandomState(0)
    x = rng.randn(2, 11, 13, cin).astype(np.float32)
    w = rng.randn(3, 3, cin, cout).astype(np.float32) * 0.3
    b = rng.randn(cout).astype(np.float32)
    y = U.conv2d_fwd(x, w, b, dil=dil, relu=True)
    xt = nhwc_to_nchw(x).requires_grad_(True)
    wt = t64(w).permute(3, 2, 0, 1).contiguous().requires_grad_(True)
    bt = t64(b).requires_grad_(True)
    pre = F.conv2d(xt, wt, bt, dilation=dil)
    yt = F.relu(pre)
    np.testing.assert_allclose(y, yt.permute(0, 2, 3, 1).detach().numpy(), **TOL)
    dy = rng.randn(*y.shape).astype(np.float32)
    yt.backward(nhwc_to_nchw(dy))
    dz = U.relu_bwd(y, dy)
    dx = U.conv2d_bwd_data(dz, w, x.shape[1:3], dil=dil)
    dw, db = U.conv2d_bwd_weight(x, dz, dil=dil)
    np.testing.assert_allclose(dx, xt.grad.permute(0, 2, 3, 1).numpy(), **TOL)
    np.testing.assert_allclose(dw, wt.grad.permute(2, 3, 1, 0).numpy(), rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(db, bt.grad.numpy(), rtol=2e-5, atol=2e-5)


def test_maxpool_fwd_bwd():
    rng = np.random.RandomState(1)
    x = rng.randn(2, 8, 6, 5).astype(np.float32)
    y = U.maxpool_fwd(x)
    xt = nhwc_to_nchw(x).requires_grad_(True)
    yt = F.max_pool2d(xt, 2, 2)
    np.testing.assert_array_equal(y, yt.permute(0, 2, 3, 1).detach().numpy().astype(np.float32))
    dy = rng.randn(*y.shape).astype(np.float32)
    yt.backward(nhwc_to_nchw(dy))
    np.testing.assert_array_equal(U.maxpool_bwd(x, dy), xt.grad.permute(0, 2, 3, 1).numpy().astype(np.float32))
    # tie-break: all-equal window -> first element of the window gets the gradient
    xz = np.zeros((1, 2, 2, 1), np.float32)
    np.testing.assert_array_equal(U.maxpool_bwd(xz, np.ones((1, 1, 1, 1), np.float32)).ravel(), [1, 0, 0, 0])


def test_convT_fwd_bwd():
    rng = np.random.RandomState(2)
    cin, cout = 6, 4
    x = rng.randn(2, 5, 7, cin).astype(np.float32)
    K = rng.randn(2, 2, cout, cin).astype(np.float32)  # TF conv2d_transpose kernel layout [kh,kw,out,in]
    b = rng.randn(cout).astype(np.float32)
    y = U.convT_fwd(x, K, b)
    xt = nhwc_to_nchw(x).requires_grad_(True)
    kt = t64(K).permute(3, 2, 0, 1).contiguous().requires_grad_(True)  # torch: [in, out, kh, kw]
    bt = t64(b).requires_grad_(True)
    yt = F.conv_transpose2d(xt, kt, bt, stride=2)
    np.testing.assert_allclose(y, yt.permute(0, 2, 3, 1).detach().numpy(), **TOL)
    dy = rng.randn(*y.shape).astype(np.float32)
    yt.backward(nhwc_to_nchw(dy))
    dx, dK, db = U.convT_bwd(x, K, dy)
    np.testing.assert_allclose(dx, xt.grad.permute(0, 2, 3, 1).numpy(), **TOL)
    np.testing.assert_allclose(dK, kt.grad.permute(2, 3, 1, 0).numpy(), rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(db, bt.grad.numpy(), rtol=2e-5, atol=2e-5)


def test_softmax_ce():
    rng = np.random.RandomState(3)
    logits = (rng.randn(2, 5, 5, 2) * 3).astype(np.float32)
    labels = (rng.rand(2, 5, 5) < 0.3).astype(np.int64)
    prob, loss, dl = U.softmax_ce(logits, labels)
    lt = t64(logits).requires_grad_(True)
    lo = F.cross_entropy(lt.reshape(-1, 2), torch.from_numpy(labels).reshape(-1))
    lo.backward()
    assert abs(loss - lo.item()) < 1e-9
    np.testing.assert_allclose(prob, torch.softmax(lt, -1)[..., 1].detach().numpy(), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(dl, lt.grad.numpy(), rtol=1e-5, atol=1e-9)


def _torch_unet(params, X, L, root, dilated):
    """the graph of the reference's unet.py:12-97 in stock torch ops, float64, NCHW (oracle/torch_ref.py)"""
    from oracle.torch_ref import torch_unet
    return torch_unet(params, X, L, root, dilated, dtype=torch.float64)


@pytest.mark.parametrize("L,root,P,dilated", [(2, 4, 12, False), (3, 4, 20, False), (3, 4, 20, True)])
def test_whole_net_loss_and_grads(L, root, P, dilated):
    S = U.input_size_needed(P, L)
    rng = np.random.RandomState(7)
    X = rng.rand(2, S, S, 3).astype(np.float32)
    labels = (rng.rand(2, P, P) < 0.2).astype(np.int64)
    params = U.init_params(L, root, dilated, seed=11, bias_scale=0.1)
    loss, probs, grads = U.loss_and_grads(params, X, labels, L, root, dilated)
    logits_t, Pt = _torch_unet(params, X, L, root, dilated)
    assert tuple(logits_t.shape) == (2, 2, P, P)
    lt = logits_t.permute(0, 2, 3, 1)
    lo = F.cross_entropy(lt.reshape(-1, 2), torch.from_numpy(labels).reshape(-1))
    lo.backward()
    assert abs(loss - lo.item()) < 1e-6
    np.testing.assert_allclose(probs, torch.softmax(lt, -1)[..., 1].detach().numpy(), rtol=1e-4, atol=1e-6)
    for name in params:
        gt = Pt[name].grad
        if gt is None:  # dead branch (deepest dilated pair): TF leaves it untouched, oracle reports zeros
            assert dilated and name.startswith("conv_dilut_%d/" % (L - 1))
            assert not grads[name].any()
            continue
        scale = max(1e-12, float(gt.abs().max()))
        err = float(np.abs(grads[name] - gt.numpy()).max()) / scale
        assert err < 5e-5, (name, err)


def test_momentum_and_lr():
    rng = np.random.RandomState(5)
    w = rng.randn(100).astype(np.float32)
    a = rng.randn(100).astype(np.float32)
    g = rng.randn(100).astype(np.float32)
    w0, a0 = w.copy(), a.copy()
    U.momentum_step(w, a, g, 0.01, 0.9)
    a_ref = np.float32(0.9) * a0 + g
    np.testing.assert_array_equal(a, a_ref)
    np.testing.assert_allclose(w, w0 - np.float32(0.01) * a_ref, rtol=1e-6, atol=1e-7)
    assert U.learning_rate(0.01, 999) == np.float32(0.01)
    assert abs(U.learning_rate(0.01, 1000) - 0.0095) < 1e-8
    assert abs(U.learning_rate(0.01, 2500) - 0.01 * 0.95 ** 2) < 1e-8


def test_param_counts():
    """report/report.tex:50 (2e8 parameters <-> L=6 dilated) and SURVEY section 8(a) counts"""
    def count(L, root, dil):
        return sum(int(np.prod(s)) for _, s in U.param_shapes(L, root, dil))
    assert count(5, 64, False) == 31031822
    assert count(6, 64, True) == 212403278
    assert count(6, 64, False) == 124362254
    assert count(3, 16, False) == 117070
