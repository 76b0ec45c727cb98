"""-m gpu: every tile shape of the second-generation conv kernel, forced (RSU_FWD2_CFG), on shapes big enough to run several
tiles per workgroup and the full-size pipeline states -- the planner alone never picks the large shapes on small test inputs.
Each case runs three times: the failures this guards against (lost accumulator lanes under register pressure)
were intermittent."""
import ctypes
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import unet_oracle as U  # noqa: E402
from road_segmentation_unet_amd._lib import RsuSrc, call, lib  # noqa: E402
from tests import hiputil as hu  # noqa: E402


# (shape, RSU_FWD_GEN): 2 = igemm_fwd2, 4 = the ping-pong kernel igemm_pp wherever it is built (shapes 0-5; 6 and 7 fall back to igemm_fwd2)
@pytest.fixture(params=[(c, g) for c in range(8) for g in (2, 4) if not (g == 4 and c >= 6)],
                ids=lambda cg: "cfg%d-gen%d" % cg)
def forced_cfg(request):
    old = {k: os.environ.get(k) for k in ("RSU_FWD2_CFG", "RSU_FWD_GEN")}
    os.environ["RSU_FWD2_CFG"] = str(request.param[0])
    os.environ["RSU_FWD_GEN"] = str(request.param[1])
    yield request.param[0]
    for k, v in old.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


def _rand(rng, *shape, scale=1.0):
    return (rng.standard_normal(shape) * scale).astype(np.float32)


@pytest.mark.parametrize("N,H,W,Cin,Cout", [(1, 150, 140, 64, 64), (1, 70, 75, 64, 192), (2, 60, 60, 16, 64)])
def test_conv3x3_fwd_and_bwd_data_every_shape(forced_cfg, N, H, W, Cin, Cout):
    rng = np.random.RandomState(H + Cout)
    x = hu.q(_rand(rng, N, H, W, Cin))
    w = _rand(rng, 3, 3, Cin, Cout, scale=1.0 / np.sqrt(9 * Cin))
    b = _rand(rng, Cout, scale=0.1)
    xd, wp, bd = hu.dev_bf16(x), hu.pack_conv_fwd(w), hu.dev_f32(b)
    ref = U.conv2d_fwd(x, hu.q(w), b)
    s = (RsuSrc * 1)(hu.src_of(xd, H, W))
    for rep in range(3):
        y = torch.full((N, H - 2, W - 2, Cout), float("nan"), dtype=torch.bfloat16, device=hu.DEV)
        call("rsu_conv2d_fwd", s, 1, hu.ptr(wp), hu.ptr(bd), hu.ptr(y), N, H, W, Cout, 1, 1, 0, hu.stream())
        hu.assert_bf16_close(hu.host(y), ref, "conv2d_fwd cfg %d rep %d" % (forced_cfg, rep))
    if Cin % 32:
        return
    dz = hu.q(_rand(rng, N, H - 2, W - 2, Cout, scale=0.1))
    dzd, wb = hu.dev_bf16(dz), hu.pack_conv_bwd(w, 0, Cin)
    rdx = U.relu_bwd(x, U.conv2d_bwd_data(dz, hu.q(w), (H, W)))
    for rep in range(3):
        dx = torch.full((N, H, W, Cin), float("nan"), dtype=torch.bfloat16, device=hu.DEV)
        call("rsu_conv2d_bwd_data", hu.ptr(dzd), hu.ptr(wb), hu.ptr(dx), hu.ptr(xd), 0, N, H, W, Cin, 0, Cin, Cout, 1, 0, hu.stream())
        hu.assert_bf16_close(hu.host(dx), rdx, "conv2d_bwd_data cfg %d rep %d" % (forced_cfg, rep))


@pytest.mark.parametrize("N,H,W,Cin,Cout", [(1, 28, 28, 256, 128), (2, 100, 100, 256, 128), (1, 60, 60, 128, 64)])
def test_convT_fwd_every_shape(forced_cfg, N, H, W, Cin, Cout):
    rng = np.random.RandomState(Cin + H)
    x = hu.q(np.maximum(_rand(rng, N, H, W, Cin), 0))
    K = _rand(rng, 2, 2, Cout, Cin, scale=1.0 / np.sqrt(Cin))
    b = _rand(rng, Cout, scale=0.1)
    pf = torch.zeros(4 * lib().rsu_packed_bytes(1, Cout, (ctypes.c_int * 1)(Cin), 1) // 2, dtype=torch.bfloat16, device=hu.DEV)
    Kd, xd, bd = hu.dev_f32(K), hu.dev_bf16(x), hu.dev_f32(b)
    call("rsu_pack_convT_fwd", hu.ptr(Kd), hu.ptr(pf), Cin, Cout, hu.stream())
    ref = U.convT_fwd(x, hu.q(K), b)
    for rep in range(3):
        y = torch.full((N, 2 * H, 2 * W, Cout), float("nan"), dtype=torch.bfloat16, device=hu.DEV)
        call("rsu_convT2x2_fwd", hu.ptr(xd), hu.ptr(pf), hu.ptr(bd), hu.ptr(y), N, H, W, Cin, Cout, 0, hu.stream())
        hu.assert_bf16_close(hu.host(y), ref, "convT fwd cfg %d rep %d" % (forced_cfg, rep))


def test_dilated_three_source_and_accumulate_every_shape(forced_cfg):
    """the remaining modes of the kernel on a multi-tile input: dilation 2, three cropped concat sources with odd channel counts,
    and the accumulating backward-data (dilated branch adding onto the main branch's input gradient)"""
    rng = np.random.RandomState(9)
    N, h, w = 1, 76, 88
    a = hu.q(_rand(rng, N, h + 12, w + 10, 48))
    bsrc = hu.q(_rand(rng, N, h + 6, w + 4, 16))
    c = hu.q(_rand(rng, N, h, w, 64))
    Cout = 128 if forced_cfg in (0, 2, 4, 6) else 64
    W = _rand(rng, 3, 3, 128, Cout, scale=0.05)
    bias = _rand(rng, Cout, scale=0.1)
    ad, bd_, cd, biasd = hu.dev_bf16(a), hu.dev_bf16(bsrc), hu.dev_bf16(c), hu.dev_f32(bias)
    wp = hu.pack_conv_fwd(W, [48, 16, 64])
    srcs = (RsuSrc * 3)(hu.src_of(ad, h, w), hu.src_of(bd_, h, w), hu.src_of(cd, h, w))
    cat = np.concatenate([U.center_crop(a, h, w), U.center_crop(bsrc, h, w), c], axis=3)
    for dil in (1, 2):
        ref = U.conv2d_fwd(cat, hu.q(W), bias, dil=dil)
        for rep in range(2):
            y = torch.full((N, h - 2 * dil, w - 2 * dil, Cout), float("nan"), dtype=torch.bfloat16, device=hu.DEV)
            call("rsu_conv2d_fwd", srcs, 3, hu.ptr(wp), hu.ptr(biasd), hu.ptr(y), N, h, w, Cout, dil, 1, 0, hu.stream())
            hu.assert_bf16_close(hu.host(y), ref, "3-source conv dil %d cfg %d" % (dil, forced_cfg))
    # accumulate: dx = base + Conv2DBackpropInput(dz) with dilation 2
    Cin = 64
    Wb = _rand(rng, 3, 3, Cin, Cout, scale=0.05)
    dz = hu.q(_rand(rng, N, h - 4, w - 4, Cout, scale=0.1))
    base = hu.q(_rand(rng, N, h, w, Cin, scale=0.1))
    dzd, wb = hu.dev_bf16(dz), hu.pack_conv_bwd(Wb, 0, Cin)
    ref = base + U.conv2d_bwd_data(dz, hu.q(Wb), (h, w), dil=2)
    for rep in range(2):
        dx = hu.dev_bf16(base).clone()
        call("rsu_conv2d_bwd_data", hu.ptr(dzd), hu.ptr(wb), hu.ptr(dx), None, 1, N, h, w, Cin, 0, Cin, Cout, 2, 0, hu.stream())
        hu.assert_bf16_close(hu.host(dx), ref, "accumulating bwd_data dil 2 cfg %d" % forced_cfg)


@pytest.mark.parametrize("wg_cfg", [0, 2])
@pytest.mark.parametrize("N,H,W,Cin,Cout,dil", [(1, 70, 75, 64, 192, 1), (2, 60, 60, 128, 128, 2)])
def test_conv3x3_bwd_weight_both_shapes(wg_cfg, N, H, W, Cin, Cout, dil):
    """weight-gradient kernel: 64x64 (two wave groups) and 128x64 workgroup shapes, multi-tile splits, three runs each"""
    old = os.environ.get("RSU_WG_CFG")
    os.environ["RSU_WG_CFG"] = str(wg_cfg)
    try:
        rng = np.random.RandomState(H + Cout + dil)
        Ho, Wo = H - 2 * dil, W - 2 * dil
        x = hu.q(_rand(rng, N, H, W, Cin))
        dz = hu.q(_rand(rng, N, Ho, Wo, Cout, scale=0.1))
        xd, dzd = hu.dev_bf16(x), hu.dev_bf16(dz)
        ref_dw, ref_db = U.conv2d_bwd_weight(x, dz, dil=dil)
        nws = lib().rsu_conv2d_bwd_weight_ws_floats(Cin, Cin, Cout)
        s = hu.src_of(xd, H, W)
        first = None
        for rep in range(3):
            ws = torch.zeros(nws + 1024, dtype=torch.float32, device=hu.DEV)
            ws[nws:] = 777.0
            dw = torch.full((3, 3, Cin, Cout), float("nan"), dtype=torch.float32, device=hu.DEV)
            db = torch.full((Cout,), float("nan"), dtype=torch.float32, device=hu.DEV)
            call("rsu_conv2d_bwd_weight", ctypes.byref(s), hu.ptr(dzd), hu.ptr(dw), hu.ptr(db), hu.ptr(ws), N, Ho, Wo, Cin, 0, Cout, dil, 0, hu.stream())
            hu.assert_f32_close(hu.host(dw), ref_dw, "bwd_weight cfg %d rep %d" % (wg_cfg, rep))
            hu.assert_f32_close(hu.host(db), ref_db, "bias grad cfg %d rep %d" % (wg_cfg, rep))
            assert bool((ws[nws:] == 777.0).all()), "workspace overrun"
            if first is None:
                first = hu.host(dw).copy()
            else:
                np.testing.assert_array_equal(hu.host(dw), first)  # fixed reduction order: bit-exact repeatability
    finally:
        if old is None:
            os.environ.pop("RSU_WG_CFG", None)
        else:
            os.environ["RSU_WG_CFG"] = old


@pytest.mark.parametrize("cfg", [0, 1, 2, 3, 4, 5])
@pytest.mark.parametrize("dil", [1, 2])
def test_pingpong_kernel_gives_the_bits_of_igemm_fwd2(cfg, dil):
    """igemm_pp (dilation 2: igemm_pp_d2.hip) keeps igemm_fwd2's summation order: forward and masked backward-data agree bit for bit, shape by shape."""
    rng = np.random.RandomState(7 + cfg)
    N, H, W, Cin, Cout = 2, 90, 84, 96, 128
    x = hu.q(_rand(rng, N, H, W, Cin))
    w = _rand(rng, 3, 3, Cin, Cout, scale=1.0 / np.sqrt(9 * Cin))
    b = _rand(rng, Cout, scale=0.1)
    dz = hu.q(_rand(rng, N, H - 2 * dil, W - 2 * dil, Cout, scale=0.1))
    xd, wp, bd, dzd, wb = hu.dev_bf16(x), hu.pack_conv_fwd(w), hu.dev_f32(b), hu.dev_bf16(dz), hu.pack_conv_bwd(w, 0, Cin)
    s = (RsuSrc * 1)(hu.src_of(xd, H, W))
    old = {k: os.environ.get(k) for k in ("RSU_FWD2_CFG", "RSU_FWD_GEN")}
    out = {}
    try:
        os.environ["RSU_FWD2_CFG"] = str(cfg)
        for gen in (2, 4):
            os.environ["RSU_FWD_GEN"] = str(gen)
            y = torch.full((N, H - 2 * dil, W - 2 * dil, Cout), float("nan"), dtype=torch.bfloat16, device=hu.DEV)
            dx = torch.full((N, H, W, Cin), float("nan"), dtype=torch.bfloat16, device=hu.DEV)
            call("rsu_conv2d_fwd", s, 1, hu.ptr(wp), hu.ptr(bd), hu.ptr(y), N, H, W, Cout, dil, 1, 0, hu.stream())
            call("rsu_conv2d_bwd_data", hu.ptr(dzd), hu.ptr(wb), hu.ptr(dx), hu.ptr(xd), 0, N, H, W, Cin, 0, Cin, Cout, dil, 0, hu.stream())
            out[gen] = (y.view(torch.int16).cpu().numpy().copy(), dx.view(torch.int16).cpu().numpy().copy())
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    assert np.array_equal(out[2][0], out[4][0]), "forward differs between igemm_fwd2 and igemm_pp (shape %d)" % cfg
    assert np.array_equal(out[2][1], out[4][1]), "backward-data differs between igemm_fwd2 and igemm_pp (shape %d)" % cfg
