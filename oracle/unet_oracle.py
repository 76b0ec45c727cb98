"""CPU oracle for the U-Net hot path: Python composition over oracle/unet_oracle.c.

TEST INFRASTRUCTURE ONLY (see the header of unet_oracle.c): imported by tests/, by
__graft_entry__.smoke() and by bench.py's cpu_baseline leg; never by the product package.

PARITY STATUS: "parity unpinned" at the TensorFlow boundary -- tensorflow==1.4.0 is the un-vendored
dependency that holds the reference arithmetic (/root/reference/requirements.txt:18-19) and it is not
installable here. This module restates, call site by call site, the graph that
/root/reference/src/unet.py:12-97 builds, the loss of src/tf_aerial_images.py:103-110,147-149 and the
optimizer of src/tf_aerial_images.py:112-122.

`emulate_bf16=True` rounds tensors to bfloat16 at exactly the points where the HIP path stores bf16
(DESIGN.md "Numerics"), so bf16 kernels can be compared tightly; with it off everything is float32
storage / float64 accumulation.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libunet_oracle.so")
_lib = None

_F = ctypes.POINTER(ctypes.c_float)
_I64 = ctypes.POINTER(ctypes.c_int64)


def build(force=False):
    """Compile oracle/unet_oracle.c with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "unet_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
    return _lib


def _fp(a):
    return a.ctypes.data_as(_F) if a is not None else None


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def round_bf16(a):
    a = _c(a).copy()
    lib().orc_round_bf16(_fp(a), ctypes.c_long(a.size))
    return a


# ----------------------------------------------------------------------------------------------
# single ops (thin wrappers; shapes NHWC, kernels HWIO)
# ----------------------------------------------------------------------------------------------
def conv1x1_fwd(x, w, b, sub=0.0):
    x, w = _c(x), _c(w)
    cin, cout = w.shape[-2], w.shape[-1]
    y = np.empty(x.shape[:-1] + (cout,), np.float32)
    lib().orc_conv1x1_fwd(_fp(x), _fp(w), _fp(_c(b)) if b is not None else None, _fp(y),
                          ctypes.c_long(x.size // cin), cin, cout, ctypes.c_float(sub))
    return y


def conv1x1_bwd(x, w, dy, sub=0.0, need_dx=True):
    x, w, dy = _c(x), _c(w), _c(dy)
    cin, cout = w.shape[-2], w.shape[-1]
    dx = np.empty_like(x) if need_dx else None
    dw = np.empty_like(w)
    db = np.empty((cout,), np.float32)
    lib().orc_conv1x1_bwd(_fp(x), _fp(w), _fp(dy), _fp(dx), _fp(dw), _fp(db), ctypes.c_long(x.size // cin), cin,
                          cout, ctypes.c_float(sub))
    return dx, dw, db


def conv2d_fwd(x, w, b, dil=1, relu=True):
    x, w = _c(x), _c(w)
    n, h, wd, cin = x.shape
    k, cout = w.shape[0], w.shape[3]
    ho, wo = h - dil * (k - 1), wd - dil * (k - 1)
    y = np.empty((n, ho, wo, cout), np.float32)
    lib().orc_conv2d_fwd(_fp(x), _fp(w), _fp(_c(b)) if b is not None else None, _fp(y), n, h, wd, cin, cout, k, dil,
                         int(relu))
    return y


def conv2d_bwd_data(dy, w, in_hw, dil=1):
    dy, w = _c(dy), _c(w)
    n = dy.shape[0]
    k, cin, cout = w.shape[0], w.shape[2], w.shape[3]
    h, wd = in_hw
    dx = np.empty((n, h, wd, cin), np.float32)
    lib().orc_conv2d_bwd_data(_fp(dy), _fp(w), _fp(dx), n, h, wd, cin, cout, k, dil)
    return dx


def conv2d_bwd_weight(x, dy, k=3, dil=1):
    x, dy = _c(x), _c(dy)
    n, h, wd, cin = x.shape
    cout = dy.shape[3]
    dw = np.empty((k, k, cin, cout), np.float32)
    db = np.empty((cout,), np.float32)
    lib().orc_conv2d_bwd_weight(_fp(x), _fp(dy), _fp(dw), _fp(db), n, h, wd, cin, cout, k, dil)
    return dw, db


def relu_bwd(y, dy):
    y, dy = _c(y), _c(dy)
    dz = np.empty_like(y)
    lib().orc_relu_bwd(_fp(y), _fp(dy), _fp(dz), ctypes.c_long(y.size))
    return dz


def maxpool_fwd(x):
    x = _c(x)
    n, h, w, c = x.shape
    y = np.empty((n, h // 2, w // 2, c), np.float32)
    lib().orc_maxpool2x2_fwd(_fp(x), _fp(y), n, h, w, c)
    return y


def maxpool_bwd(x, dy):
    x, dy = _c(x), _c(dy)
    n, h, w, c = x.shape
    dx = np.empty_like(x)
    lib().orc_maxpool2x2_bwd(_fp(x), _fp(dy), _fp(dx), n, h, w, c)
    return dx


def convT_fwd(x, K, b):
    x, K = _c(x), _c(K)
    n, h, w, cin = x.shape
    cout = K.shape[2]
    y = np.empty((n, 2 * h, 2 * w, cout), np.float32)
    lib().orc_convT2x2s2_fwd(_fp(x), _fp(K), _fp(_c(b)) if b is not None else None, _fp(y), n, h, w, cin, cout)
    return y


def convT_bwd(x, K, dy, need_dx=True):
    x, K, dy = _c(x), _c(K), _c(dy)
    n, h, w, cin = x.shape
    cout = K.shape[2]
    dx = np.empty_like(x) if need_dx else None
    dK = np.empty_like(K)
    db = np.empty((cout,), np.float32)
    lib().orc_convT2x2s2_bwd(_fp(x), _fp(K), _fp(dy), _fp(dx), _fp(dK), _fp(db), n, h, w, cin, cout)
    return dx, dK, db


def softmax_ce(logits, labels=None, want_grad=True):
    logits = _c(logits)
    npix = logits.size // 2
    prob = np.empty(logits.shape[:-1], np.float32)
    loss = ctypes.c_double(0.0)
    if labels is None:
        lib().orc_softmax_ce(_fp(logits), None, _fp(prob), None, None, ctypes.c_long(npix))
        return prob, None, None
    labels = np.ascontiguousarray(labels, dtype=np.int64)
    dl = np.empty_like(logits) if want_grad else None
    lib().orc_softmax_ce(_fp(logits), labels.ctypes.data_as(_I64), _fp(prob), ctypes.byref(loss), _fp(dl),
                         ctypes.c_long(npix))
    return prob, float(loss.value), dl


def momentum_step(w, acc, g, lr, mu):
    """In place on float32 contiguous arrays (tf_aerial_images.py:116-121)."""
    assert w.dtype == np.float32 and acc.dtype == np.float32 and w.flags.c_contiguous and acc.flags.c_contiguous
    g = _c(g)
    lib().orc_momentum_step(_fp(w), _fp(acc), _fp(g), ctypes.c_float(lr), ctypes.c_float(mu), ctypes.c_long(w.size))


def center_crop(t, h, w):
    """tf.image.resize_image_with_crop_or_pad when the target is smaller: offset floor((H-h)/2) (unet.py:70-83)."""
    oy, ox = (t.shape[1] - h) // 2, (t.shape[2] - w) // 2
    return t[:, oy:oy + h, ox:ox + w, :]


def center_pad_like(d, full_shape):
    """adjoint of center_crop"""
    out = np.zeros(full_shape, np.float32)
    h, w = d.shape[1], d.shape[2]
    oy, ox = (full_shape[1] - h) // 2, (full_shape[2] - w) // 2
    out[:, oy:oy + h, ox:ox + w, :] = d
    return out


# ----------------------------------------------------------------------------------------------
# geometry + parameters
# ----------------------------------------------------------------------------------------------
def input_size_needed(output_size, num_layers):
    """unet.py:100-115 restated (closed form S = P + 12*2^(L-1) - 8 for valid P)."""
    o = output_size
    for i in range(num_layers - 1):
        assert o % 2 == 0, 'expand layer {} has size {} not divisible by 2'.format(num_layers - i, o)
        o = (o + 4) / 2
    for i in range(num_layers - 1):
        o = (o + 4) * 2
    return int(o + 4)


def param_shapes(num_layers, root_size, dilated_layers):
    """Variable set created by unet.forward (unet.py:23,34-45,67,88-91,95), TF names, TF layouts, creation order."""
    shapes = [("color_space_adjust/kernel", (1, 1, 3, 3)), ("color_space_adjust/bias", (3,))]
    nf, cin = root_size, 3
    for i in range(num_layers):
        if dilated_layers:
            shapes += [("conv_dilut_%d/atrous_conv1/kernel" % i, (3, 3, cin, nf)),
                       ("conv_dilut_%d/atrous_conv1/bias" % i, (nf,)),
                       ("conv_dilut_%d/atrous_conv2/kernel" % i, (3, 3, nf, nf)),
                       ("conv_dilut_%d/atrous_conv2/bias" % i, (nf,))]
        shapes += [("conv_%d/conv1/kernel" % i, (3, 3, cin, nf)), ("conv_%d/conv1/bias" % i, (nf,)),
                   ("conv_%d/conv2/kernel" % i, (3, 3, nf, nf)), ("conv_%d/conv2/bias" % i, (nf,))]
        cin = nf
        nf *= 2
    nf //= 2
    for i in range(num_layers - 1):
        nf //= 2
        shapes += [("up_conv_%d/kernel" % i, (2, 2, nf, 2 * nf)), ("up_conv_%d/bias" % i, (nf,))]
        ccat = (3 if dilated_layers else 2) * nf
        j = num_layers + i
        shapes += [("conv_%d/conv1/kernel" % j, (3, 3, ccat, nf)), ("conv_%d/conv1/bias" % j, (nf,)),
                   ("conv_%d/conv2/kernel" % j, (3, 3, nf, nf)), ("conv_%d/conv2/bias" % j, (nf,))]
    shapes += [("weight_output/kernel", (1, 1, nf, 2)), ("weight_output/bias", (2,))]
    return shapes


def init_params(num_layers, root_size, dilated_layers, seed=2018, bias_scale=0.0):
    """Glorot-uniform kernels, zero biases (tf.layers defaults). bias_scale>0 gives non-zero biases for tests."""
    rng = np.random.RandomState(seed)
    params = {}
    for name, shp in param_shapes(num_layers, root_size, dilated_layers):
        if name.endswith("kernel"):
            recept = shp[0] * shp[1]
            limit = np.sqrt(6.0 / (recept * shp[2] + recept * shp[3]))
            params[name] = rng.uniform(-limit, limit, size=shp).astype(np.float32)
        else:
            params[name] = (bias_scale * rng.standard_normal(shp)).astype(np.float32)
    return params


# ----------------------------------------------------------------------------------------------
# whole network
# ----------------------------------------------------------------------------------------------
def dropout_key(seed, site, step):
    """32-bit key of dropout site `site` (encoder level i -> i, decoder stage i -> L + i) at optimizer step `step`"""
    return (int(seed) * 0x9E3779B9 + int(site) * 0x85EBCA6B + int(step) * 0xC2B2AE35) & 0xFFFFFFFF


def dropout_mask(shape, keep, key):
    """floor(keep + U) of tf.nn.dropout (unet.py:29-30,64-65) with U = hash(key, NHWC element index) / 2^24 -- the counter-based
    stand-in for TF's Philox stream that the HIP kernels use (TF's own stream cannot be reproduced). float32 0/1 array."""
    n = int(np.prod(shape))
    assert n <= 0xFFFFFFFF
    with np.errstate(over="ignore"):
        h = np.arange(n, dtype=np.uint32) ^ np.uint32(key)
        h ^= h >> np.uint32(16)
        h *= np.uint32(0x85EBCA6B)
        h ^= h >> np.uint32(13)
        h *= np.uint32(0xC2B2AE35)
        h ^= h >> np.uint32(16)
    u = (h >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    return np.floor(np.float32(keep) + u).astype(np.float32).reshape(shape)


def _q(t, emu):
    return round_bf16(t) if emu else t


def forward(params, X, num_layers, root_size, dilated_layers, emulate_bf16=False, keep_cache=True, keep=1.0, seed=0, step=0):
    """unet.forward (unet.py:12-97). keep = dropout_keep (1.0: identity); (seed, step) select the dropout masks.
    Returns (logits, cache)."""
    emu = emulate_bf16
    inv_keep = np.float32(1.0) / np.float32(keep)

    def drop(t, site):  # x / keep * floor(keep + U), unet.py:30,65
        if keep >= 1.0:
            return None
        return dropout_mask(t.shape, keep, dropout_key(seed, site, step)) * inv_keep
    q = lambda t: _q(t, emu)  # noqa: E731
    qw = lambda name: _q(params[name], emu)  # MFMA kernels read bf16 copies of the weights  # noqa: E731
    cache = {}
    X = _c(X)
    # unet.py:22-23
    net = conv1x1_fwd(X, params["color_space_adjust/kernel"][0, 0], params["color_space_adjust/bias"], sub=0.5)
    skips = []
    for i in range(num_layers):
        mk = drop(net, i)
        cache["mk_%d" % i] = mk
        inp = q(net * mk) if mk is not None else q(net)  # (level 0: the one bf16 rounding of net0; pooled tensors are bf16 already)
        cache["in_%d" % i] = inp
        first = False  # (level-0 conv1 also runs on the MFMA kernel with bf16 weight copies)
        dil_out = None
        last = (i == num_layers - 1)
        if dilated_layers and not last:  # the level L-1 dilated pair is dead code (unet.py:57-59)
            w1 = params["conv_dilut_%d/atrous_conv1/kernel" % i] if first else qw("conv_dilut_%d/atrous_conv1/kernel" % i)
            d1 = q(conv2d_fwd(inp, w1, params["conv_dilut_%d/atrous_conv1/bias" % i], dil=2))
            d2 = q(conv2d_fwd(d1, qw("conv_dilut_%d/atrous_conv2/kernel" % i),
                              params["conv_dilut_%d/atrous_conv2/bias" % i], dil=2))
            cache["dil1_%d" % i], cache["dil2_%d" % i] = d1, d2
            dil_out = d2
        w1 = params["conv_%d/conv1/kernel" % i] if first else qw("conv_%d/conv1/kernel" % i)
        c1 = q(conv2d_fwd(inp, w1, params["conv_%d/conv1/bias" % i]))
        c2 = q(conv2d_fwd(c1, qw("conv_%d/conv2/kernel" % i), params["conv_%d/conv2/bias" % i]))
        cache["c1_%d" % i], cache["c2_%d" % i] = c1, c2
        skips.append((c2, dil_out))
        if not last:  # the deepest pool is dead (unet.py:56)
            net = maxpool_fwd(c2)
    net = skips.pop()[0]
    for i in range(num_layers - 1):
        j = num_layers + i
        cache["uporig_%d" % i] = net
        mk = drop(net, num_layers + i)
        cache["mk_%d" % (num_layers + i)] = mk
        if mk is not None:
            net = q(net * mk)
        cache["upin_%d" % i] = net
        up = q(convT_fwd(net, qw("up_conv_%d/kernel" % i), params["up_conv_%d/bias" % i]))
        skip, dskip = skips.pop()
        parts = [center_crop(skip, up.shape[1], up.shape[2])]
        if dilated_layers:
            parts.append(center_crop(dskip, up.shape[1], up.shape[2]))
        parts.append(up)
        cat = np.ascontiguousarray(np.concatenate(parts, axis=3))  # unet.py:79/85 order [skip,(dil skip),up]
        cache["cat_%d" % i] = cat
        c1 = q(conv2d_fwd(cat, qw("conv_%d/conv1/kernel" % j), params["conv_%d/conv1/bias" % j]))
        c2 = q(conv2d_fwd(c1, qw("conv_%d/conv2/kernel" % j), params["conv_%d/conv2/bias" % j]))
        cache["c1_%d" % j], cache["c2_%d" % j] = c1, c2
        net = c2
    assert not skips
    cache["last"] = net
    logits = conv1x1_fwd(net, params["weight_output/kernel"][0, 0], params["weight_output/bias"])  # unet.py:95
    return logits, (cache if keep_cache else None)


def predict_probs(params, X, num_layers, root_size, dilated_layers, emulate_bf16=False):
    """tf_aerial_images.py:147-148: softmax(logits)[..., 1]"""
    logits, _ = forward(params, X, num_layers, root_size, dilated_layers, emulate_bf16, keep_cache=False)
    return softmax_ce(logits)[0]


def loss_and_grads(params, X, labels, num_layers, root_size, dilated_layers, emulate_bf16=False, keep=1.0, seed=0, step=0):
    """Forward + mean sparse softmax CE (tf_aerial_images.py:103-110) + full backward.
    Returns (loss, probs, grads dict keyed by TF variable name)."""
    emu = emulate_bf16
    q = lambda t: _q(t, emu)  # noqa: E731
    qw = lambda name: _q(params[name], emu)  # noqa: E731
    L = num_layers
    logits, c = forward(params, X, L, root_size, dilated_layers, emu, keep=keep, seed=seed, step=step)
    probs, loss, dlogits = softmax_ce(logits, labels)
    g = {}
    last = c["last"]
    dact, dw, db = conv1x1_bwd(last, params["weight_output/kernel"][0, 0], dlogits)
    g["weight_output/kernel"] = dw.reshape(1, 1, *dw.shape)
    g["weight_output/bias"] = db
    dz = q(relu_bwd(last, dact))  # gradient wrt pre-activation of the last conv2, stored bf16 in the HIP path

    def conv_block_bwd(name1, name2, x_in, y1, y2, dz2, dil, first=False, need_dx=True):
        """backward through relu(conv2(relu(conv1(x_in)))); dz2 = grad wrt conv2 pre-activation."""
        dw2, db2 = conv2d_bwd_weight(y1, dz2, dil=dil)
        g[name2 + "/kernel"], g[name2 + "/bias"] = dw2, db2
        dy1 = conv2d_bwd_data(dz2, qw(name2 + "/kernel"), y1.shape[1:3], dil=dil)
        dz1 = q(relu_bwd(y1, dy1))
        dw1, db1 = conv2d_bwd_weight(x_in, dz1, dil=dil)
        g[name1 + "/kernel"], g[name1 + "/bias"] = dw1, db1
        if not need_dx:
            return None, dz1
        w1 = params[name1 + "/kernel"] if first else qw(name1 + "/kernel")
        return conv2d_bwd_data(dz1, w1, x_in.shape[1:3], dil=dil), dz1

    # decoder, deepest-last in forward => walk i = L-2 .. 0
    dskips = {}
    for i in reversed(range(L - 1)):
        j = L + i
        cat = c["cat_%d" % i]
        dcat, _ = conv_block_bwd("conv_%d/conv1" % j, "conv_%d/conv2" % j, cat, c["c1_%d" % j], c["c2_%d" % j], dz, 1)
        lvl = L - 2 - i  # encoder level whose skip this decoder stage consumed
        nf = c["c2_%d" % j].shape[3]
        dskip_main = q(dcat[..., :nf])
        if dilated_layers:
            dskip_dil = q(dcat[..., nf:2 * nf])
            dup = q(dcat[..., 2 * nf:])
        else:
            dskip_dil = None
            dup = q(dcat[..., nf:])
        dskips[lvl] = (dskip_main, dskip_dil)
        upin = c["upin_%d" % i]
        dupin, dK, dbk = convT_bwd(upin, qw("up_conv_%d/kernel" % i), dup)
        g["up_conv_%d/kernel" % i], g["up_conv_%d/bias" % i] = dK, dbk
        # the transposed conv's input is (the dropped copy of) the ReLU output of the previous block's conv2
        mk = c["mk_%d" % j]
        if mk is not None:
            dupin = dupin * mk
        if i > 0:
            dz = q(relu_bwd(c["uporig_%d" % i], dupin))
        else:
            dz_bottom = q(relu_bwd(c["uporig_%d" % i], dupin))

    # encoder, level L-1 .. 0
    dpool = None  # gradient wrt the pooled tensor feeding level i+1
    for i in reversed(range(L)):
        y2 = c["c2_%d" % i]
        if i == L - 1:
            dz2 = dz_bottom if L > 1 else dz
        else:
            mk = c["mk_%d" % (i + 1)]  # the pooled tensor went through the next level's dropout
            gsum = maxpool_bwd(y2, dpool * mk if mk is not None else dpool) + center_pad_like(dskips[i][0], y2.shape)
            dz2 = q(relu_bwd(y2, gsum))
        first = (i == 0)
        dx_main, _ = conv_block_bwd("conv_%d/conv1" % i, "conv_%d/conv2" % i, c["in_%d" % i], c["c1_%d" % i], y2, dz2,
                                    1, first=first, need_dx=True)
        dx = dx_main
        if dilated_layers and i < L - 1:
            d2 = c["dil2_%d" % i]
            dzd2 = q(relu_bwd(d2, center_pad_like(dskips[i][1], d2.shape)))
            dx_dil, _ = conv_block_bwd("conv_dilut_%d/atrous_conv1" % i, "conv_dilut_%d/atrous_conv2" % i,
                                       c["in_%d" % i], c["dil1_%d" % i], d2, dzd2, 2, first=first, need_dx=True)
            # HIP path: the main branch's bwd-data stores bf16, the dilated branch's accumulates onto it
            dx = (q(dx_main) if i > 0 else dx_main) + dx_dil
        elif dilated_layers:  # dead level: parameters exist, receive no gradient (tf.gradients returns None -> untouched)
            for nm in ("atrous_conv1", "atrous_conv2"):
                g["conv_dilut_%d/%s/kernel" % (i, nm)] = np.zeros_like(params["conv_dilut_%d/%s/kernel" % (i, nm)])
                g["conv_dilut_%d/%s/bias" % (i, nm)] = np.zeros_like(params["conv_dilut_%d/%s/bias" % (i, nm)])
        dpool = q(dx) if i > 0 else dx
    # color_space_adjust: dnet0 = dpool (gradient wrt net0, fp32 here)
    if c["mk_0"] is not None:
        dpool = _c(dpool * c["mk_0"])
    _, dw0, db0 = conv1x1_bwd(_c(X), params["color_space_adjust/kernel"][0, 0], dpool, sub=0.5, need_dx=False)
    g["color_space_adjust/kernel"] = dw0.reshape(1, 1, 3, 3)
    g["color_space_adjust/bias"] = db0
    return loss, probs, g


def learning_rate(lr0, global_step):
    """tf.train.exponential_decay(lr, step, 1000, 0.95, staircase=True) (tf_aerial_images.py:116-117)"""
    return np.float32(lr0) * np.float32(0.95) ** np.float32(global_step // 1000)


def train_step(params, accums, X, labels, num_layers, root_size, dilated_layers, lr0=0.01, momentum=0.9,
               global_step=0, emulate_bf16=False, keep=1.0, seed=0):
    """One session.run of [train, loss, predictions] (tf_aerial_images.py:241-244). Updates params/accums in place."""
    loss, probs, grads = loss_and_grads(params, X, labels, num_layers, root_size, dilated_layers, emulate_bf16, keep=keep, seed=seed,
                                        step=global_step)
    lr = float(learning_rate(lr0, global_step))
    for name in params:
        momentum_step(params[name], accums[name], grads[name], lr, momentum)
    return loss, probs, grads
