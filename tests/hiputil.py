"""Helpers for the -m gpu parity tests: move numpy data to the device as bf16/f32, call the C ABI, compare."""
import ctypes

import numpy as np
import torch

from oracle import unet_oracle as U
from road_segmentation_unet_amd._lib import RsuSrc, call, lib

DEV = "cuda:0"
BF16_ULP = 2.0 ** -7  # worst-case relative spacing of bfloat16


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def dev_bf16(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV).to(torch.bfloat16).contiguous()


def dev_f32(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV).contiguous()


def host(t):
    torch.cuda.synchronize()
    return t.detach().to(torch.float32).cpu().numpy()


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def q(a):
    """bf16 round trip on the host (same RNE as the device)"""
    return U.round_bf16(np.asarray(a, dtype=np.float32))


def src_of(t, h, w, oy=None, ox=None):
    H, W, C = t.shape[1], t.shape[2], t.shape[3]
    return RsuSrc(t.data_ptr(), H, W, C, (H - h) // 2 if oy is None else oy, (W - w) // 2 if ox is None else ox)


def pack_conv_fwd(w, segs=None):
    k, cin, cout = w.shape[0], w.shape[2], w.shape[3]
    segs = segs or [cin]
    seg = (ctypes.c_int * len(segs))(*segs)
    nbytes = lib().rsu_packed_bytes(k * k, cout, seg, len(segs))
    out = torch.zeros(nbytes // 2, dtype=torch.bfloat16, device=DEV)
    wd = dev_f32(w)
    call("rsu_pack_conv_fwd", ptr(wd), ptr(out), k, cin, cout, seg, len(segs), stream())
    return out


def pack_conv_bwd(w, ci_off=0, ci_cnt=None):
    k, cin, cout = w.shape[0], w.shape[2], w.shape[3]
    ci_cnt = ci_cnt or cin
    seg = (ctypes.c_int * 1)(cout)
    out = torch.zeros(lib().rsu_packed_bytes(k * k, ci_cnt, seg, 1) // 2, dtype=torch.bfloat16, device=DEV)
    wd = dev_f32(w)
    call("rsu_pack_conv_bwd", ptr(wd), ptr(out), k, cin, ci_off, ci_cnt, cout, stream())
    return out


def assert_bf16_close(got, ref, what, ulps=1.0, atol_scale=2e-5):
    """got: device result that was rounded to bf16 once; ref: oracle result (float32, same bf16-rounded inputs).
    Allowed: `ulps` bf16 ulp of relative error (rounding-boundary flips from a different fp32 summation order)
    plus atol_scale * max|ref| (fp32 accumulation noise where the result cancels)."""
    got = np.asarray(got, np.float32)
    refq = q(ref)
    assert got.shape == refq.shape, (what, got.shape, refq.shape)
    scale = float(np.abs(refq).max()) if refq.size else 1.0
    tol = ulps * BF16_ULP * np.abs(refq) + atol_scale * max(scale, 1e-30)
    err = np.abs(got - refq)
    bad = err > tol
    if bad.any():
        idx = np.unravel_index(np.argmax(err - tol), err.shape)
        raise AssertionError("%s: %d/%d elements out of tolerance; worst at %s got %r ref %r (max|ref| %g); exact-match frac %.4f"
                             % (what, int(bad.sum()), bad.size, idx, got[idx], refq[idx], scale, float((got == refq).mean())))
    return float((got == refq).mean())


def assert_f32_close(got, ref, what, rtol=1e-4, atol_scale=1e-5):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    scale = float(np.abs(ref).max()) if ref.size else 1.0
    tol = rtol * np.abs(ref) + atol_scale * max(scale, 1e-30)
    err = np.abs(got - ref)
    if (err > tol).any():
        idx = np.unravel_index(np.argmax(err - tol), err.shape)
        raise AssertionError("%s: %d/%d out of tolerance; worst at %s got %r ref %r (max|ref| %g)"
                             % (what, int((err > tol).sum()), err.size, idx, got[idx], ref[idx], scale))
