"""Host-side mirror of the reference's network definition (/root/reference/src/unet.py) over the HIP C ABI.

`input_size_needed` and `forward` keep the reference's names, argument meaning and error behaviour
(unet.py:12, unet.py:100); `UNet` owns what the TF graph + session owned: the variables (TF names, TF
layouts, float32), their Momentum slots, the activation buffers and the backward pass
(tf_aerial_images.py:103-122). PyTorch is only the device-memory / stream / collective plumbing: every
arithmetic kernel on the path is a hand-written HIP kernel reached through librsu_hip.so.
"""
import ctypes
import math
import os

import numpy as np
import torch

from . import _lib
from ._lib import RsuSrc, RsuWgradJob, call

NUM_CHANNELS = 3  # src/constants.py:3
NUM_LABELS = 2    # src/constants.py:4


# RSU_WG_GROUP: how the weight gradients of a backward pass are launched (UNet._flush_wgrads). Default: one launch per layer where a
# side stream runs them beside the backward-data launches (each kernel on half the chip: per layer they are as efficient there as a
# group, and the fine grain keeps both streams busy to the end -- measured, profiles/r03/wg_group_schedules.txt), ONE grouped launch behind
# the pass where everything runs on one stream (alone on the chip a layer's launch pays 75 MB of slabs and a partly filled last round)
_WG_GROUP_TWO_STREAMS, _WG_GROUP_ONE_STREAM = "0", "all"
# RSU_RAW_EVENTS=0: fork the side stream through torch events (system-scope release) instead of _lib.hip_fork (agent scope)
_RAW_EVENTS = os.environ.get("RSU_RAW_EVENTS", "1") != "0"
_NO_FORK_PROBE = os.environ.get("RSU_NO_FORK_PROBE", "0") == "1"   # developer timing probe (profiles/r06/fork_bound.txt): NO dependency at all
_WG_EVENTS = os.environ.get("RSU_WG_EVENTS", "0") == "1"   # grouped weight gradients: a torch event per queued job (rounds 2-3)
_SPLIT_DEFAULT = "128,128"   # RSU_SPLIT_CHIP: CUs the main stream / each side stream plan for during the backward pass (UNet._Side)


def input_size_needed(output_size, num_layers):
    """Utility function to compute image size for a given U-Net output (reference: unet.py:100-115).

    Same arithmetic, same AssertionError text as the reference."""
    for i in range(num_layers - 1):
        assert output_size % 2 == 0, 'expand layer {} has size {} not divisible by 2' \
            .format(num_layers - i, output_size)
        output_size = (output_size + 4) / 2
    for i in range(num_layers - 1):
        output_size = (output_size + 4) * 2
    return int(output_size + 4)


def param_shapes(num_layers, root_size, dilated_layers):
    """(name, shape) of every variable unet.forward creates, in creation order (unet.py:23,34-45,67,88-91,95)."""
    shapes = [("color_space_adjust/kernel", (1, 1, 3, 3)), ("color_space_adjust/bias", (3,))]
    nf, cin = root_size, NUM_CHANNELS
    for i in range(num_layers):
        if dilated_layers:
            shapes += [("conv_dilut_%d/atrous_conv1/kernel" % i, (3, 3, cin, nf)), ("conv_dilut_%d/atrous_conv1/bias" % i, (nf,)),
                       ("conv_dilut_%d/atrous_conv2/kernel" % i, (3, 3, nf, nf)), ("conv_dilut_%d/atrous_conv2/bias" % i, (nf,))]
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
    shapes += [("weight_output/kernel", (1, 1, nf, NUM_LABELS)), ("weight_output/bias", (NUM_LABELS,))]
    return shapes


def _is_dead(name, num_layers):
    """The level L-1 dilated pair is built but never consumed (unet.py:57-59): no gradient, never updated."""
    return name.startswith("conv_dilut_%d/" % (num_layers - 1))


def glorot_uniform_params(num_layers, root_size, dilated_layers, seed):
    """tf.layers defaults: glorot_uniform kernels, zero biases. (TF's own seeds are op-id derived and not
    reproducible outside TF; parity tests inject identical weights instead.)"""
    rng = np.random.RandomState(seed)
    out = {}
    for name, shp in param_shapes(num_layers, root_size, dilated_layers):
        if name.endswith("kernel"):
            recept = shp[0] * shp[1]
            limit = math.sqrt(6.0 / (recept * shp[2] + recept * shp[3]))
            out[name] = rng.uniform(-limit, limit, size=shp).astype(np.float32)
        else:
            out[name] = np.zeros(shp, np.float32)
    return out


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def cu_shares(full, parts):
    """CUs the backward streams plan for: `parts` = RSU_SPLIT_CHIP as integers ([main, side, ...] out of 256), `full` = the CUs the backward pass
    may use in all (256, or a data-parallel budget that leaves CUs to RCCL's channel workgroups). With the default 128 + 128 and a budget of
    224 .. 255 the weight-gradient stream keeps its 128 -- its pixel splits and workgroup counts per XCD are powers of two: 120 + 120 costs c2
    8 % and c4 5 %, 112 + 128 costs 1.7 % and 3.6 % (profiles/r06/dp_budget.txt) -- and the backward-data stream, whose persistent kernels walk
    tile lists of any length, takes the rest. Below 224 the backward-data stream would starve (80 + 128: -16 %): shares in proportion, in
    steps of 8, at least 32 -- as for every other setting."""
    if full < 256 and list(parts) == [128, 128] and full >= 224:
        return [full - 128, 128]
    return [max(32, v * full // 256 // 8 * 8) for v in parts]


def _src(t, h, w):
    """window (h, w) centred in NHWC tensor t (crop offset floor((H-h)/2), unet.py:70-83)"""
    H, W, C = t.shape[1], t.shape[2], t.shape[3]
    return RsuSrc(t.data_ptr(), H, W, C, (H - h) // 2, (W - w) // 2)


class UNet:
    """Variables + buffers + forward/backward of the U-Net for a FIXED (batch, patch_size) -- the reference's graph is
    static in exactly the same way (tf_aerial_images.py:133-138)."""

    def __init__(self, num_layers, root_size, dilated_layers, batch_size, patch_size, device="cuda:0", params=None, seed=2017,
                 training=True):
        assert root_size % 8 == 0 and (root_size // 8) & (root_size // 8 - 1) == 0, "root_size must be 8 * 2^k for the HIP path"
        self.L, self.root, self.dilated = num_layers, root_size, bool(dilated_layers)
        self.B, self.P = batch_size, patch_size
        self.S = input_size_needed(patch_size, num_layers)
        self.device = torch.device(device)
        if self.device.type == "cuda":
            torch.cuda.set_device(self.device)  # the library works on the HIP current device (rsu.h "devices")
        self.training = training
        self.keep = 1.0        # dropout keep probability of the forward pass in flight (set by forward_device)
        self.dropout_seed = int(seed) if seed is not None else 0
        self.wstream = None    # side stream for the weight-gradient launches (see _Side); wstreams: all of them
        self.wstreams = []
        self._split = None     # (full, main, [side ...]) CU budgets while a backward pass shares the chip between the streams
        self.backward_cu_budget = None   # CUs the backward launches may plan for in total (None: the library's default)
        self._ncu = 0          # `ncu` argument of the MFMA launches issued now (0: the library's default budget)
        self._tuned = set()    # (training?, backward_cu_budget, dropout?) combinations the tuning pass has run for (tune / ensure_tuned)
        # grouped weight gradients (rsu.h rsu_wgrad_group_*): the launches of RSU_WG_GROUP consecutive levels / decoder stages of the
        # backward pass go out as ONE launch (0: one launch per layer, as in round 2; "all": one group behind the whole pass)
        self._wg_sizes, self._wg_group = [0], 0   # set per backward pass (_wg_policy)
        self._wg_pending, self._wgT_pending, self._wg_levels, self._wg_event, self._wg_index = [], [], 0, None, 0
        self._side_active = False   # a weight-gradient launch has gone to the side stream in this backward pass
        self._wg_plans = {}
        self._side_rr = 0
        self._side_dirty = True
        if training and self.device.type == "cuda" and os.environ.get("RSU_WGRAD_STREAM", "1") == "1":
            nside = max(1, len(os.environ.get("RSU_SPLIT_CHIP", _SPLIT_DEFAULT).split(",")) - 1)
            self.wstreams = [torch.cuda.Stream(device=self.device) for _ in range(nside)]
            self.wstream = self.wstreams[0]
        # Momentum + re-pack fused into the weight-gradient side (backward_device(update=...), world size 1): the backward-data launches of
        # step t read set `_pkset` of the backward-data packs while the fused updates of the same step write the other set
        self._pkset, self.pk_alt, self._fused_tabs, self._fused, self._fused_pending = 0, {}, {}, None, None
        self._update_tables, self._update_table = {}, None
        self.pool_code = {}
        self.prof = None       # list collecting (tag, algorithmic flops, start event, end event, CU share) when profiling
        self.on_grads = None   # callback(lo): every gradient at flat position >= lo is final (see dist.GradBucketer)
        _lib.lib()  # fail loudly now if the HIP extension is missing
        self._check_tensor_sizes()
        self._alloc_params(params if params is not None else glorot_uniform_params(num_layers, root_size, dilated_layers, seed))
        self._alloc_buffers()
        self.repack()

    # ------------------------------------------------------------------ parameters
    def _alloc_params(self, init):
        shapes = param_shapes(self.L, self.root, self.dilated)
        live = [(n, s) for n, s in shapes if not _is_dead(n, self.L)]
        dead = [(n, s) for n, s in shapes if _is_dead(n, self.L)]
        self.names = [n for n, _ in shapes]
        off, self._slices = 0, {}
        for n, s in live + dead:
            if n == (dead[0][0] if dead else None):
                self.n_live = off
            cnt = int(np.prod(s))
            self._slices[n] = (off, cnt, s)
            off += (cnt + 3) // 4 * 4  # 16-byte aligned starts
        if not dead:
            self.n_live = off
        self.n_flat = off
        dev = self.device
        self.flat_w = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_acc = torch.zeros(off, dtype=torch.float32, device=dev)   # Momentum slots (tf_aerial_images.py:120)
        self.flat_g = torch.zeros(off, dtype=torch.float32, device=dev)
        self.w, self.acc, self.g = {}, {}, {}
        for n, (o, cnt, s) in self._slices.items():
            self.w[n] = self.flat_w[o:o + cnt].view(s)
            self.acc[n] = self.flat_acc[o:o + cnt].view(s)
            self.g[n] = self.flat_g[o:o + cnt].view(s)
            self.w[n].copy_(torch.from_numpy(np.ascontiguousarray(init[n], dtype=np.float32)))
        self.global_step = 0

    def state_dict(self):
        """TF variable names, TF layouts (HWIO / [kh,kw,out,in]) + Momentum slots + global_step."""
        d = {n: self.w[n].detach().cpu().numpy().copy() for n in self.names}
        d.update({n + "/Momentum": self.acc[n].detach().cpu().numpy().copy() for n in self.names})
        d["global_step"] = np.int64(self.global_step)
        return d

    def load_state_dict(self, d):
        self._load_count = getattr(self, "_load_count", 0) + 1
        for n in self.names:
            self.w[n].copy_(torch.from_numpy(np.ascontiguousarray(d[n], dtype=np.float32)))
            if n + "/Momentum" in d:
                self.acc[n].copy_(torch.from_numpy(np.ascontiguousarray(d[n + "/Momentum"], dtype=np.float32)))
        self.global_step = int(d.get("global_step", 0))
        self.repack()

    def _check_tensor_sizes(self):
        """The kernels address a tensor through 32-bit byte offsets: every activation must stay below 2 GiB (the entry points
        return RSU_E2BIG otherwise). Checked here, where the batch is chosen, with the largest batch that fits."""
        h0 = self.S - 2                                    # level-0 conv1 output, `root` channels: the largest tensor of a level
        per_patch = max(self.S * self.S * 16, h0 * h0 * self.root) * 2
        limit = 0x7ffffff0
        if self.B * per_patch >= limit:
            raise _lib.RsuError("batch_size %d is too large for num_layers=%d patch_size=%d: the level-0 activations (%d bytes per patch) "
                                "must stay below 2 GiB per tensor; the largest per-GPU batch is %d"
                                % (self.B, self.L, self.P, per_patch, (limit - 1) // per_patch))

    # ------------------------------------------------------------------ buffers
    def _bf(self, *shape):
        return torch.zeros(shape, dtype=torch.bfloat16, device=self.device)

    def _alloc_buffers(self):
        B, L, S = self.B, self.L, self.S
        dev = self.device
        self.x = torch.zeros((B, S, S, 3), dtype=torch.float32, device=dev)
        self.in16 = self._bf(B, S, S, 16)
        self.act, self.grad = {}, {}
        h, nf = S, self.root
        self.level_in = {}
        for i in range(L):
            self.level_in[i] = h
            if self.dilated and i < L - 1:
                self.act["d1_%d" % i] = self._bf(B, h - 4, h - 4, nf)
                self.act["d2_%d" % i] = self._bf(B, h - 8, h - 8, nf)
            self.act["c1_%d" % i] = self._bf(B, h - 2, h - 2, nf)
            self.act["c2_%d" % i] = self._bf(B, h - 4, h - 4, nf)
            if i < L - 1:
                assert (h - 4) % 2 == 0
                self.act["pool_%d" % i] = self._bf(B, (h - 4) // 2, (h - 4) // 2, nf)
                h = (h - 4) // 2
                nf *= 2
        h = h - 4
        for i in range(L - 1):
            nf //= 2
            j = L + i
            self.act["up_%d" % i] = self._bf(B, 2 * h, 2 * h, nf)
            h = 2 * h
            self.act["c1_%d" % j] = self._bf(B, h - 2, h - 2, nf)
            self.act["c2_%d" % j] = self._bf(B, h - 4, h - 4, nf)
            h -= 4
        assert h == self.P, (h, self.P)
        self.last_name = "c2_%d" % (2 * L - 2) if L > 1 else "c2_0"
        self.prob = torch.zeros((B, self.P, self.P), dtype=torch.float32, device=dev)
        self.logits = torch.zeros((B, self.P, self.P, 2), dtype=torch.float32, device=dev)
        self.labels = torch.zeros((B, self.P, self.P), dtype=torch.int64, device=dev)
        self.loss_sum = torch.zeros(1, dtype=torch.float32, device=dev)
        # workspace of the conv launches that cut their reduction into slices (rsu.h rsu_conv2d_fwd_k: the deep levels at small batches);
        # one per stream that issues conv launches -- the main stream, and the side stream of the dilated twin blocks in the forward pass
        nk = int(_lib.lib().rsu_conv_splitk_ws_floats()) if os.environ.get("RSU_KSPLIT", "1") != "0" else 0
        self.kws = torch.zeros(nk, dtype=torch.float32, device=dev) if nk else None
        # (one per side stream: whether a layer splits must never depend on which stream its launch went to)
        self.kws_side = [torch.zeros(nk, dtype=torch.float32, device=dev) for _ in self.wstreams] if (nk and self.dilated) else []
        if self.training:
            for k, t in self.act.items():
                if k.startswith("up_") or k.startswith("c") or k.startswith("d") or k.startswith("pool_"):
                    self.grad[k] = torch.zeros_like(t)
            # cropped skip gradients (decoder conv1 bwd-data outputs for the skip sources)
            for i in range(L - 1):
                up = self.act["up_%d" % i]
                self.grad["skip_%d" % i] = torch.zeros_like(up)
                if self.dilated:
                    self.grad["skipd_%d" % i] = torch.zeros_like(up)
            lib = _lib.lib()
            ws = [lib.rsu_head_ws_floats(B * self.P * self.P, self.root), lib.rsu_conv_first_bwd_ws_floats(self.root)]
            for n, (_, _, s) in self._slices.items():
                if n.endswith("kernel") and len(s) == 4 and s[0] == 3 and s[2] != 3:
                    srcs = self._conv_sources_c(n, s)
                    for c in srcs:
                        ws.append(lib.rsu_conv2d_bwd_weight_ws_floats(s[2], c, s[3]))
                    ws.append(lib.rsu_bias_grad_ws_floats(B * S * S, s[3]))
                if n.startswith("up_conv") and n.endswith("kernel"):
                    ws.append(lib.rsu_convT2x2_bwd_weight_ws_floats(s[3], s[2]))
            ws.append(lib.rsu_wgrad_group_ws_floats())
            self.ws = torch.zeros(int(max(ws)) + 64, dtype=torch.float32, device=dev)
            # (one workspace per stream that launches weight gradients: their slabs are live at the same time)
            self.ws_side = [self.ws] + [torch.zeros_like(self.ws) for _ in self.wstreams[1:]]
            self.gfirst = torch.zeros((2, 9, 12, self.root), dtype=torch.float32, device=dev)  # gx of conv1 / atrous_conv1 (rsu.h)
            # code bytes of the max-pools (argmax + ReLU bits per pooled element, rsu_maxpool2x2_fwd_code): the gradient junction of a
            # level reads them instead of the level's conv2 activation. RSU_POOL_CODE=0: it reads the activation (same bits)
            if os.environ.get("RSU_POOL_CODE", "1") == "1":
                for i in range(L - 1):
                    self.pool_code[i] = torch.zeros(self.act["pool_%d" % i].shape, dtype=torch.uint8, device=dev)

    def _conv_sources_c(self, name, shape):
        """channel counts of the concat sources feeding conv `name` (decoder conv1: [skip,(dil skip),up], unet.py:79/85)"""
        cin, cout = shape[2], shape[3]
        blk = int(name.split("/")[0].split("_")[-1]) if name.startswith("conv_") and not name.startswith("conv_dilut") else -1
        if name.endswith("conv1/kernel") and blk >= self.L:
            return [cout] * (3 if self.dilated else 2)
        return [cin]

    # ------------------------------------------------------------------ packed bf16 weights
    def repack(self):
        """float32 master weights -> bf16 MFMA fragment order (after init / load / every optimizer step): ONE kernel launch
        over a device-resident job table (rsu_pack_table_*) built on first use."""
        lib = _lib.lib()
        st = ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream) if self.device.type == "cuda" else None
        if not hasattr(self, "pk"):
            self.pk = {}
            jobs = []  # (kind, weight name, packed tensor, k, Cin_total, ci_off, ci_cnt, Cout, segs)

            def buf(nbytes):
                return torch.zeros(nbytes // 2, dtype=torch.bfloat16, device=self.device)
            for n, (_, _, s) in self._slices.items():
                if not n.endswith("kernel") or _is_dead(n, self.L):
                    continue
                if n.startswith("up_conv"):
                    cout, cin = s[2], s[3]
                    self.pk[n, "fwd"] = buf(4 * lib.rsu_packed_bytes(1, cout, (ctypes.c_int * 1)(cin), 1))
                    self.pk[n, "bwd"] = buf(lib.rsu_packed_bytes(4, cin, (ctypes.c_int * 1)(cout), 1))
                    jobs.append((2, n, self.pk[n, "fwd"], 2, cin, 0, cin, cout, None))
                    if self.training:
                        jobs.append((3, n, self.pk[n, "bwd"], 2, cin, 0, cin, cout, None))
                elif s[0] == 3 and s[2] == NUM_CHANNELS:  # level-0 conv1 / atrous_conv1 over the 16-channel input tensor
                    self.pk[n, "fwd"] = buf(lib.rsu_packed_first_bytes(s[3]))
                    jobs.append((4, n, self.pk[n, "fwd"], 3, 3, 0, 3, s[3], None))
                elif s[0] == 3:
                    cin, cout = s[2], s[3]
                    segs = self._conv_sources_c(n, s)
                    self.pk[n, "fwd"] = buf(lib.rsu_packed_bytes(9, cout, (ctypes.c_int * len(segs))(*segs), len(segs)))
                    jobs.append((0, n, self.pk[n, "fwd"], 3, cin, 0, cin, cout, segs))
                    off = 0
                    for si, c in enumerate(segs):  # one backward-data pack per concat source
                        self.pk[n, "bwd", si] = buf(lib.rsu_packed_bytes(9, c, (ctypes.c_int * 1)(cout), 1))
                        if self.training:
                            jobs.append((1, n, self.pk[n, "bwd", si], 3, cin, off, c, cout, None))
                        off += c
            esz = lib.rsu_pack_table_entry_bytes()
            host = ctypes.create_string_buffer(esz * (len(jobs) + 3 * sum(1 for j in jobs if j[0] == 2)))
            idx = 0
            for kind, n, packed, k, cin, off, cnt, cout, segs in jobs:
                seg = (ctypes.c_int * len(segs))(*segs) if segs else None
                used = lib.rsu_pack_table_add(host, idx, kind, _ptr(self.w[n]), _ptr(packed), k, cin, off, cnt, cout, seg, len(segs) if segs else 0)
                if used < 1:
                    raise _lib.RsuError("rsu_pack_table_add(%s) failed: %d" % (n, used))
                idx += used
            nb = ctypes.c_int(0)
            _lib.check(lib.rsu_pack_table_finish(host, idx, ctypes.byref(nb)), "rsu_pack_table_finish")
            self._pack_n, self._pack_blocks = idx, nb.value
            self._pack_table = torch.frombuffer(bytearray(host.raw[:esz * idx]), dtype=torch.uint8).to(self.device)
        call("rsu_pack_table_run", _ptr(self._pack_table), self._pack_n, self._pack_blocks, st)
        self._pkset = 0   # (the pack table writes set 0 of the backward-data packs)

    # ------------------------------------------------------------------ forward
    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _timed(self, tag, flops, fn, *args):
        """call an ABI entry point; when profiling, bracket it with events on the launch stream"""
        if self.prof is None:
            call(fn, *args)
            return
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        call(fn, *args)
        e1.record()
        # (the share of the chip this launch plans for: bench.py prices a launch's duration by it -- two half-chip launches side by side)
        self.prof.append((tag, flops, e0, e1, (self._ncu or _lib.lib().rsu_get_cu_budget()) / 256.0))

    def _grads_ready(self, name):
        if self.on_grads is not None:  # (dist.GradBucketer waits for the weight-gradient stream itself: extra_streams)
            self._flush_wgrads()       # queued weight gradients must be ON the side stream before the exchange may wait for it
            self.on_grads(self._slices[name][0])

    # Weight-gradient launches can go to a second stream: they only READ what the main stream produced (dz, activations) and
    # write gradients nobody reads before the optimizer. Two persistent kernels then share the chip and each fills the other's
    # poorly occupied last round of tiles. RSU_WGRAD_STREAM=0 keeps everything on one stream.
    #
    # The two streams SHARE the chip by plan (RSU_SPLIT_CHIP, default _SPLIT_DEFAULT, of 256): while the backward pass runs, launches on
    # the main stream (backward-data) plan their persistent workgroups for the first number of CUs, launches on the side stream
    # (weight gradients) for the second, so a backward-data and a weight-gradient kernel are resident together, each on its
    # own CUs, instead of taking turns on all of them. The weight gradient of a layer is a sum over one partial result PER
    # WORKGROUP (a 295-KB slab each, written and read back by the reduce kernel): half the workgroups, half that traffic --
    # and persistent kernels on fewer CUs lose less to their last, partly filled round of tiles. "0" = both plan for every CU.
    def _wg_policy(self):
        """RSU_WG_GROUP for this backward pass: "n" every group holds n blocks; "a,b,c" the first group a blocks, the second b, ... (the
        last number repeats); "all" one group behind the whole pass; "0" one launch per layer"""
        g = os.environ.get("RSU_WG_GROUP", _WG_GROUP_TWO_STREAMS if self.wstreams else _WG_GROUP_ONE_STREAM)
        self._wg_sizes = [10 ** 6] if g == "all" else [max(0, int(v)) for v in g.split(",")]
        self._wg_group = self._wg_sizes[0]

    def _begin_split(self):
        """CU shares of the backward pass: `backward_cu_budget` (set by the data-parallel host: CUs left to RCCL's channel workgroups
        while the gradient exchange overlaps the backward pass; the forward pass keeps the whole chip) shared out between the
        streams. Every launch carries its share as its own `ncu` argument (self._ncu): no library state changes between launches."""
        self._split = None
        self._wg_policy()
        full = self.backward_cu_budget or _lib.lib().rsu_get_cu_budget()
        spec = os.environ.get("RSU_SPLIT_CHIP", _SPLIT_DEFAULT)
        parts = None
        if self.wstreams and spec not in ("0", "") and self._wg_group < 10 ** 6:   # (RSU_WG_GROUP=all: nothing runs beside backward-data)
            try:
                parts = [int(v) for v in spec.split(",")]
            except ValueError:
                parts = None
            if parts is not None and (len(parts) != len(self.wstreams) + 1 or min(parts) < 1):
                parts = None
        if parts is None:
            self._ncu = full if self.backward_cu_budget else 0
            return
        parts = cu_shares(full, parts)
        self._split = (full, parts[0], parts[1:])
        # the main stream keeps the whole budget until the first weight-gradient launch has gone to the side stream
        self._ncu = full if self.backward_cu_budget else 0

    def _end_split(self):
        self._split = None
        self._ncu = 0
        self._side_active = False
        self._wg_index = 0

    class _Side:
        """`with UNet._Side(net) as side:` -- launches inside go to the next side stream (round robin) and plan for that stream's share
        of the chip; side.ws is its workspace"""

        def __init__(self, net, alone=False, after=None):
            self.net, self.ctx, self.alone = net, None, alone  # alone: nothing is left to run beside it on the main stream
            self.after = after   # event on the main stream behind which the side stream may start (None: everything issued so far)
            self.ws = net.ws if net.training else None
            self.saved_ncu = net._ncu

        def __enter__(self):
            n = self.net
            self.saved_ncu = n._ncu
            if not n.wstreams:
                return self
            k = n._side_rr % len(n.wstreams)
            n._side_rr += 1
            n._side_dirty = True
            self.ws = n.ws_side[k]
            if n._split is not None:
                n._ncu = n._split[0] if self.alone else n._split[2][k]
            ev = self.after
            if _NO_FORK_PROBE and n._split is not None and n._side_active:
                # (timing probe only: behind the first fork of a backward pass the side stream no longer waits for the main stream -- wrong
                # numbers, the bound of a dependency that costs neither queue anything: profiles/r06/fork_bound.txt)
                pass
            elif ev is None and _RAW_EVENTS:
                # the fork costs the MAIN queue an idle gap per weight-gradient launch (the event's packet sits between two backward-data
                # kernels): ~6 us with a torch event, less without the system-scope fence a same-device dependency does not need
                try:
                    _lib.hip_fork(torch.cuda.current_stream(n.device).cuda_stream, n.wstreams[k].cuda_stream, n.device.index)
                except (_lib.RsuError, OSError, AttributeError) as ex:   # no usable runtime handle: torch's events do the same, a little slower
                    import sys
                    print("road_segmentation_unet_amd: raw HIP fork events unavailable (%r); using torch events" % (ex,), file=sys.stderr)
                    globals()["_RAW_EVENTS"] = False
                    ev = torch.cuda.Event()
                    ev.record(torch.cuda.current_stream(n.device))
                    n.wstreams[k].wait_event(ev)
            else:
                if ev is None:
                    ev = torch.cuda.Event()
                    ev.record(torch.cuda.current_stream(n.device))
                n.wstreams[k].wait_event(ev)
            self.ctx = torch.cuda.stream(n.wstreams[k])
            self.ctx.__enter__()
            return self

        def __exit__(self, *a):
            try:
                if self.ctx is not None:
                    self.ctx.__exit__(*a)
            finally:
                n = self.net
                n._ncu = self.saved_ncu   # on error paths too
                if n._split is not None and self.ctx is not None and not n._side_active:
                    n._side_active = True   # from here on the main stream's launches share the chip with the side stream's
                    n._ncu = n._split[1]

    def _join_side(self):
        if not self._side_dirty:   # nothing has gone to a side stream since the last join: no wait packet on the main queue
            return
        for s in self.wstreams:
            torch.cuda.current_stream(self.device).wait_stream(s)
        self._side_dirty = False

    def _conv(self, name, srcs, hin, out, dil=1):
        arr = (RsuSrc * len(srcs))(*srcs)
        cout = out.shape[3]
        cin = sum(s.C for s in srcs)
        ho = hin - 2 * dil
        kws = self._kws()
        nb = out.shape[0]   # (the batch of the tensors at hand: forward_device may run the two halves of a batch as two chains)
        self._timed("conv3x3_fwd", 2.0 * nb * ho * ho * cout * cin * 9, "rsu_conv2d_fwd_k", arr, len(srcs),
                    _ptr(self.pk[name + "/kernel", "fwd"]), _ptr(self.w[name + "/bias"]), _ptr(out), nb, hin, hin, cout, dil, 1,
                    self._ncu, _ptr(kws), kws.numel() if kws is not None else 0, self._stream())

    def _kws(self):
        """the split-K workspace of the stream the next conv launch goes to (None: that launch never splits)"""
        if self.kws is None or not self.wstreams:
            return self.kws
        cur = torch.cuda.current_stream(self.device)
        for k, s in enumerate(self.wstreams):
            if cur == s:
                return self.kws_side[k] if k < len(self.kws_side) else None
        return self.kws

    def _conv_first(self, name, out, h, dil, st, in16=None, x=None):
        if x is not None:
            # colour adjust + first conv in one launch, straight from the f32 input (rsu.h rsu_color_conv_first_fwd; same bits, no dropout)
            call("rsu_color_conv_first_fwd", _ptr(x), _ptr(self.w["color_space_adjust/kernel"]), _ptr(self.w["color_space_adjust/bias"]),
                 _ptr(self.pk[name + "/kernel", "fwd"]), _ptr(self.w[name + "/bias"]), _ptr(out), out.shape[0], h, h, self.root, dil, self._ncu, st)
            return
        call("rsu_conv_first_fwd", _ptr(self.in16 if in16 is None else in16), _ptr(self.pk[name + "/kernel", "fwd"]), _ptr(self.w[name + "/bias"]),
             _ptr(out), out.shape[0], h, h, self.root, dil, self._ncu, st)

    def dropout_key(self, site):
        """32-bit key of dropout site `site` (encoder level i -> i, decoder stage i -> L + i: the 2L-1 tf.nn.dropout calls of
        unet.py:29-30,64-65) for the current global step; the hash of (key, element index) replaces TF's Philox stream"""
        return (self.dropout_seed * 0x9E3779B9 + site * 0x85EBCA6B + self.global_step * 0xC2B2AE35) & 0xFFFFFFFF

    def forward_device(self, want_logits=False, keep=1.0):
        """unet.forward (unet.py:12-97) on self.x (already on device); keep = dropout_keep (1.0: none); fills self.prob."""
        keep = float(keep)
        assert 0.0 < keep <= 1.0, "dropout keep probability must be in (0, 1]"
        self.keep = keep
        for f in self._forward_steps(0, self.B, want_logits, keep):
            f()
        return self.act[self.last_name]

    def _forward_steps(self, b0, nb, want_logits, keep):
        """the launches of a forward pass over images [b0, b0 + nb) as a list of closures (one ABI call each, issued on the stream that is
        current when the closure runs). (Two chains over the halves of a batch on two streams were measured: -2..-4 % on the step,
        profiles/r04/fwd_halves_probe.txt.)"""
        L, S = self.L, self.S
        a = {k: t[b0:b0 + nb] for k, t in self.act.items()}
        x, in16, prob, logits = self.x[b0:b0 + nb], self.in16[b0:b0 + nb], self.prob[b0:b0 + nb], self.logits[b0:b0 + nb]
        code = {k: t[b0:b0 + nb] for k, t in self.pool_code.items()}
        steps = []

        def add(f):
            steps.append(f)
        # A forward-only net without dropout reads the f32 input in the level-0 convs themselves (rsu_color_conv_first_fwd: one launch, the
        # 16-channel tensor is never built, same bits). A training step needs that tensor for the first conv's weight gradient anyway, and the
        # first conv reads it back hot from the Infinity Cache: measured (profiles/r06/kernel_stats_fused_first.md), the fused launch beside a
        # colour-adjust launch on the side stream takes 66 us against 14.5 + 48 us of the two launches in a row -- training keeps the two.
        # RSU_FIRST_FUSED=0: never fuse; =2: fuse in training steps too (the measured variant).
        ff = os.environ.get("RSU_FIRST_FUSED", "1")
        fused_first = keep == 1.0 and (ff == "2" or (ff != "0" and not self.training))
        xin = x if fused_first else None

        def color_adjust():
            call("rsu_color_adjust_fwd", _ptr(x), _ptr(self.w["color_space_adjust/kernel"]), _ptr(self.w["color_space_adjust/bias"]),
                 _ptr(in16), nb * S * S, keep, self.dropout_key(0), self._stream())
        if not fused_first:
            add(color_adjust)
        elif self.training:
            if len(self.wstreams) == 1:   # (one side stream: its later launches that read in16 follow in order; the main stream joins it before its own)
                def color_adjust_side():
                    with UNet._Side(self):
                        color_adjust()
                add(color_adjust_side)
            else:
                add(color_adjust)
        cur, h = None, S
        for i in range(L):
            last = i == L - 1
            if self.dilated and not last:
                # the dilated twin block (unet.py:32-39) reads the level's input and feeds only a decoder skip: it runs on the second
                # stream beside the main block (joined in front of the decoder)
                def twin(i=i, cur=cur, h=h):
                    with UNet._Side(self):
                        if i == 0:
                            self._conv_first("conv_dilut_0/atrous_conv1", a["d1_0"], h, 2, self._stream(), in16, xin)
                        else:
                            self._conv("conv_dilut_%d/atrous_conv1" % i, [_src(cur, h, h)], h, a["d1_%d" % i], dil=2)
                        self._conv("conv_dilut_%d/atrous_conv2" % i, [_src(a["d1_%d" % i], h - 4, h - 4)], h - 4, a["d2_%d" % i], dil=2)
                add(twin)
            if i == 0:
                add(lambda h=h: self._conv_first("conv_0/conv1", a["c1_0"], h, 1, self._stream(), in16, xin))
            else:
                add(lambda i=i, cur=cur, h=h: self._conv("conv_%d/conv1" % i, [_src(cur, h, h)], h, a["c1_%d" % i]))
            if last:
                add(lambda i=i, h=h: self._conv("conv_%d/conv2" % i, [_src(a["c1_%d" % i], h - 2, h - 2)], h - 2, a["c2_%d" % i]))
            else:
                # conv2 + ReLU + the level's 2x2 max-pool (+ the next level's dropout, + the code bytes of the gradient junction): one
                # call, one launch where the pool folds into the conv's epilogue (rsu.h rsu_conv2d_fwd_pool)
                def conv2_pool(i=i, h=h):
                    c1, c2 = a["c1_%d" % i], a["c2_%d" % i]
                    nf_ = c2.shape[3]
                    src = (RsuSrc * 1)(_src(c1, h - 2, h - 2))
                    kws = self._kws()
                    self._timed("conv3x3_fwd", 2.0 * nb * (h - 4) * (h - 4) * nf_ * c1.shape[3] * 9, "rsu_conv2d_fwd_pool_k", src, 1,
                                _ptr(self.pk["conv_%d/conv2/kernel" % i, "fwd"]), _ptr(self.w["conv_%d/conv2/bias" % i]), _ptr(c2),
                                _ptr(a["pool_%d" % i]), _ptr(code.get(i)), nb, h - 2, h - 2, nf_, keep, self.dropout_key(i + 1), self._ncu,
                                _ptr(kws), kws.numel() if kws is not None else 0, self._stream())
                add(conv2_pool)
                cur, h = a["pool_%d" % i], (h - 4) // 2
        net, h = a["c2_%d" % (L - 1)], h - 4
        if self.dilated:
            add(self._join_side)
        for i in range(L - 1):
            j, lvl = L + i, L - 2 - i
            up = a["up_%d" % i]
            if keep < 1.0:  # unet.py:64-65
                if self.act.get("drop_%d" % i) is None:
                    self.act["drop_%d" % i] = torch.zeros_like(self.act["c2_%d" % (j - 1)] if i > 0 else self.act["c2_%d" % (L - 1)])
                drop = self.act["drop_%d" % i][b0:b0 + nb]
                add(lambda net=net, drop=drop, i=i: call("rsu_dropout_fwd", _ptr(net), _ptr(drop), net.numel(), keep, self.dropout_key(L + i), self._stream()))
                net = drop
            add(lambda net=net, up=up, i=i, h=h: call("rsu_convT2x2_fwd", _ptr(net), _ptr(self.pk["up_conv_%d/kernel" % i, "fwd"]),
                                                     _ptr(self.w["up_conv_%d/bias" % i]), _ptr(up), nb, h, h, net.shape[3], up.shape[3], self._ncu, self._stream()))
            h = 2 * h

            def dec(j=j, lvl=lvl, up=up, h=h):
                srcs = [_src(a["c2_%d" % lvl], h, h)]
                if self.dilated:
                    srcs.append(_src(a["d2_%d" % lvl], h, h))
                srcs.append(_src(up, h, h))
                self._conv("conv_%d/conv1" % j, srcs, h, a["c1_%d" % j])
            add(dec)
            add(lambda j=j, h=h: self._conv("conv_%d/conv2" % j, [_src(a["c1_%d" % j], h - 2, h - 2)], h - 2, a["c2_%d" % j]))
            net, h = a["c2_%d" % j], h - 4
        if not self.training or want_logits:
            add(lambda net=net: call("rsu_head_fwd", _ptr(net), _ptr(self.w["weight_output/kernel"]), _ptr(self.w["weight_output/bias"]), _ptr(prob),
                                     _ptr(logits) if want_logits else None, nb * self.P * self.P, self.root, self._stream()))
        return steps

    # ------------------------------------------------------------------ backward
    def _bwd_pack(self, kname, si, other=False):
        """the backward-data pack of concat source si of conv kernel `kname` in the set the backward-data launches read now (other: the
        set the fused updates of this step write; allocated on first use)"""
        if (self._pkset == 0) != other:
            return self.pk[kname, "bwd", si]
        return self._bwd_pack_set1(kname, si)

    def _fused_names(self):
        """kernels whose Momentum step rides on their weight-gradient launches (backward_device(update=...)): every 3x3 conv kernel the
        MFMA weight-gradient kernels serve -- all but the level-0 conv1 kernels (Cin = 3) and the transposed convs"""
        return [n for n, (o, c, sh) in self._slices.items()
                if o < self.n_live and n.endswith("kernel") and len(sh) == 4 and sh[0] == 3 and sh[2] != NUM_CHANNELS and not n.startswith("up_conv")]

    def _fused_entry(self, kname):
        """host pointer to the rsu_update_table_add entry of `kname` whose backward-data packs are the set NOT read in this step"""
        tab = self._fused_tabs.get(kname)
        lib = _lib.lib()
        esz = lib.rsu_update_table_entry_bytes()
        if tab is None:
            sh = self._slices[kname][2]
            segs = self._conv_sources_c(kname, sh)
            host = ctypes.create_string_buffer(esz * 2)
            keep = []
            for st_ in (0, 1):   # entry st_ WRITES set st_
                ptrs = [(self.pk[kname, "bwd", si] if st_ == 0 else self._bwd_pack_set1(kname, si)).data_ptr() for si in range(len(segs))]
                bw = (ctypes.c_void_p * len(segs))(*ptrs)
                keep.append(bw)
                rc = lib.rsu_update_table_add(host, st_, 0, _ptr(self.w[kname]), _ptr(self.acc[kname]), _ptr(self.g[kname]), _ptr(self.pk[kname, "fwd"]), bw,
                                              sh[2], sh[3], (ctypes.c_int * len(segs))(*segs), len(segs))
                if rc != 1:
                    raise _lib.RsuError("rsu_update_table_add(%s) failed: %d" % (kname, rc))
            tab = self._fused_tabs[kname] = (host, keep)
        return ctypes.c_void_p(ctypes.addressof(tab[0]) + esz * (1 - self._pkset))

    def _bwd_pack_set1(self, kname, si):
        t = self.pk_alt.get((kname, si))
        if t is None:
            t = self.pk_alt[kname, si] = torch.zeros_like(self.pk[kname, "bwd", si])
        return t

    def _wgrad(self, name, srcs_t, dz, hout, dil=1):
        """dW (HWIO rows per source) + db of conv `name`; srcs_t = list of (tensor, window size). With grouping on (RSU_WG_GROUP) the
        launches are only queued here; _flush_wgrads sends a whole group to the side stream as one launch."""
        cout = dz.shape[3]
        cin_total = self.w[name + "/kernel"].shape[2]
        off = 0
        if self._wg_group > 0:
            for t, win in srcs_t:
                db = self.g[name + "/bias"] if off == 0 else None  # BiasAddGrad rides along with the first source's launch
                job = RsuWgradJob(_lib.WGRAD_CONV3X3, _src(t, win, win), dz.data_ptr(), self.g[name + "/kernel"].data_ptr(),
                                  db.data_ptr() if db is not None else None, hout, hout, cin_total, off, cout, dil)
                self._wg_pending.append((job, 2.0 * self.B * hout * hout * cout * t.shape[3] * 9, (name, off)))
                off += t.shape[3]
            # (rounds 2-3 recorded a torch event on the main stream here, per queued job, so that a group flushed later would not wait for
            # more than it reads: every record costs the main queue ~5 us between two backward-data kernels -- RSU_WG_EVENTS=1 restores it.
            # Now the group forks where it is flushed, through the raw event of _Side.)
            if _WG_EVENTS:
                self._wg_event = self._record_main()
            if name.endswith("conv2") and not (name.startswith("conv_0/") and not self.dilated):   # (the last group: flushed by the caller, which knows that nothing runs beside it)
                # a group closes BEHIND a conv2 gradient: dz of a block's conv2 is there when the block's backward pass begins, so the
                # group {conv1 (+ transposed conv) of the block before, conv2 of this one} can run beside ALL of this block's
                # backward-data launches; closing it behind conv1 would make it wait for the block's first backward-data launch
                self._wg_levels += 1
                if self._wg_levels >= self._wg_sizes[min(self._wg_index, len(self._wg_sizes) - 1)]:
                    self._flush_wgrads()
            return
        with UNet._Side(self) as side:
            st = self._stream()
            for si, (t, win) in enumerate(srcs_t):
                s = _src(t, win, win)
                db = _ptr(self.g[name + "/bias"]) if off == 0 else None  # BiasAddGrad rides along with the first source's launch
                fl = 2.0 * self.B * hout * hout * cout * t.shape[3] * 9
                if self._fused is not None:
                    # the launch that sums the slabs is the Momentum step + re-pack of the rows this source owns (rsu.h)
                    lr, mu, keep_grad = self._fused
                    self._timed("conv3x3_bwd_weight", fl, "rsu_conv2d_bwd_weight_update", ctypes.byref(s), _ptr(dz), _ptr(self.g[name + "/kernel"]), db,
                                _ptr(side.ws), self.B, hout, hout, cin_total, off, cout, dil, self._ncu, self._fused_entry(name + "/kernel"), si, lr, mu, 1.0,
                                keep_grad, st)
                else:
                    self._timed("conv3x3_bwd_weight", fl, "rsu_conv2d_bwd_weight", ctypes.byref(s), _ptr(dz), _ptr(self.g[name + "/kernel"]), db,
                                _ptr(side.ws), self.B, hout, hout, cin_total, off, cout, dil, self._ncu, st)
                off += t.shape[3]

    def _wgradT(self, i, upin, dup, hh, nf):
        """dK + db of up_conv_i (queued like _wgrad when grouping is on)"""
        if self._wg_group > 0:
            job = RsuWgradJob(_lib.WGRAD_CONVT2X2, RsuSrc(upin.data_ptr(), hh, hh, upin.shape[3], 0, 0), dup.data_ptr(),
                              self.g["up_conv_%d/kernel" % i].data_ptr(), self.g["up_conv_%d/bias" % i].data_ptr(), 0, 0, 0, 0, nf, 1)
            self._wgT_pending.append((job, 0.0, ("up_conv_%d" % i, upin.data_ptr())))
            if _WG_EVENTS:
                self._wg_event = self._record_main()
            return
        with UNet._Side(self) as side:
            call("rsu_convT2x2_bwd_weight", _ptr(upin), _ptr(dup), _ptr(self.g["up_conv_%d/kernel" % i]), _ptr(self.g["up_conv_%d/bias" % i]),
                 _ptr(side.ws), self.B, hh, hh, upin.shape[3], nf, self._ncu, self._stream())

    def _record_main(self):
        if not self.wstreams:
            return None
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        return ev

    def _flush_wgrads(self, alone=False):
        """Launch the queued weight gradients as grouped launches on the side stream (everything they read has been produced by launches
        issued on the main stream before this point; _Side makes the side stream wait for them). The plan of a group -- which layers,
        which CU share, which workspace -- is made once and kept (host table + its device copy, rsu.h rsu_wgrad_group_plan)."""
        if not self._wg_pending and not self._wgT_pending:
            return   # (the data-parallel host asks per bucket boundary: an empty flush must not advance the group schedule)
        self._wg_levels = 0
        self._wg_index += 1
        for pending, tag in ((self._wg_pending, "conv3x3_bwd_weight"), (self._wgT_pending, None)):
            while pending:
                chunk, rest = pending[:_lib.WGRAD_GROUP_MAX], pending[_lib.WGRAD_GROUP_MAX:]
                del pending[:]
                pending.extend(rest)
                with UNet._Side(self, alone=alone, after=self._wg_event if _WG_EVENTS else None) as side:
                    key = (tuple(k for _, _, k in chunk), self._ncu, side.ws.data_ptr())
                    plan = self._wg_plans.get(key)
                    if plan is None:
                        lib = _lib.lib()
                        nb = lib.rsu_wgrad_group_table_bytes()
                        host = ctypes.create_string_buffer(nb)
                        arr = (RsuWgradJob * len(chunk))(*[j for j, _, _ in chunk])
                        _lib.check(lib.rsu_wgrad_group_plan(arr, len(chunk), _ptr(side.ws), self.B, self._ncu, host), "rsu_wgrad_group_plan")
                        devt = torch.frombuffer(bytearray(host.raw), dtype=torch.uint8).to(self.device)
                        plan = self._wg_plans[key] = (host, devt)
                    flops = sum(f for _, f, _ in chunk)
                    if tag is None:
                        call("rsu_wgrad_group_run", plan[0], _ptr(plan[1]), self._stream())
                    else:
                        self._timed(tag, flops, "rsu_wgrad_group_run", plan[0], _ptr(plan[1]), self._stream())

    def _bwd_data(self, name, dz, dx, hin, relu_src=None, accumulate=0, src_index=0, dil=1):
        """Conv2DBackpropInput towards concat source `src_index` of conv `name` (its own weight pack)"""
        cout = self.w[name + "/kernel"].shape[3]
        cnt = dx.shape[3]
        ho = hin - 2 * dil
        kws = self._kws()
        self._timed("conv3x3_bwd_data", 2.0 * self.B * ho * ho * cout * cnt * 9, "rsu_conv2d_bwd_data_k", _ptr(dz),
                    _ptr(self._bwd_pack(name + "/kernel", src_index)), _ptr(dx), _ptr(relu_src), accumulate, self.B, hin, hin, cnt, 0, cnt,
                    cout, dil, self._ncu, _ptr(kws), kws.numel() if kws is not None else 0, self._stream())

    def backward_device(self, inv_count, update=None, keep_grad=False):
        """loss + all gradients for self.x / self.labels; forward_device() must have run. inv_count = 1 / (global pixel count).
        update = (lr0, momentum): single-device training -- the MomentumOptimizer step (tf_aerial_images.py:120-121) and the re-pack of every
        3x3 conv kernel ride on the kernel's weight-gradient launches (rsu.h rsu_conv2d_bwd_weight_update: the gradient is neither written
        nor re-read, the pass leaves the tail of the step); apply_momentum(lr0, momentum) MUST follow and then only steps the remaining
        variables. Same bits as the plain pair of calls. keep_grad: self.g of those kernels is written as well (it is not otherwise).
        OPT-IN (RSU_FUSED_WGRAD=1; otherwise, and where the weight gradients are grouped or a gradient exchange runs, the plain schedule
        runs and apply_momentum steps everything): measured in round 6, the pass costs the weight-gradient stream -- which is as long as
        the backward-data stream -- more than it saves behind the pass: c2 955 -> 888 patches/s, c3 212 -> 184 (profiles/r06/abenv_fused2_*.txt;
        an HBM-bound pass on the half of the chip the weight-gradient kernel leaves it, beside a backward-data kernel)."""
        B, L, st, a, g = self.B, self.L, self._stream(), self.act, self.grad
        keep = self.keep
        last = a[self.last_name]
        self._begin_split()
        self._fused = self._fused_pending = None
        if (update is not None and self.training and self._wg_group == 0 and self.on_grads is None and self.device.type == "cuda"
                and os.environ.get("RSU_FUSED_WGRAD", "0") == "1" and os.environ.get("RSU_FUSED_UPDATE", "1") != "0"):
            self._fused = (self.learning_rate(update[0]), float(update[1]), 1 if keep_grad else 0)
            self._fused_pending = (float(update[0]), float(update[1]))
        try:
            self._backward_body(inv_count)
        finally:
            self._fused = None
            self._end_split()   # (an exception inside must not leave later launches planned for a share of the chip)
        # ---- color_space_adjust (unet.py:22-23): its input gradient is never materialised (include/rsu.h, rsu_conv_first_bwd_weight):
        #   dW0[ci][cj] = 1/keep * sum_{t,co} W1[t][cj][co] * gxc[t][ci][cj][co];  db0[cj] = 1/keep * sum_{t,co} W1[t][cj][co] * gm[t][cj][co]
        inv_keep = float(np.float32(1.0) / np.float32(keep))
        gk, gb = self.g["color_space_adjust/kernel"], self.g["color_space_adjust/bias"]
        call("rsu_color_adjust_bwd", _ptr(self.gfirst[0]), _ptr(self.w["conv_0/conv1/kernel"]), _ptr(gk), _ptr(gb), self.root, inv_keep, 0, st)
        if self.dilated and L > 1:
            call("rsu_color_adjust_bwd", _ptr(self.gfirst[1]), _ptr(self.w["conv_dilut_0/atrous_conv1/kernel"]), _ptr(gk), _ptr(gb), self.root,
                 inv_keep, 1, st)

    def _backward_body(self, inv_count):
        B, L, st, a, g = self.B, self.L, self._stream(), self.act, self.grad
        keep = self.keep
        last = a[self.last_name]
        self.loss_sum.zero_()
        call("rsu_head_fwd_bwd", _ptr(last), _ptr(self.w["weight_output/kernel"]), _ptr(self.w["weight_output/bias"]), _ptr(self.labels),
             _ptr(self.prob), _ptr(self.loss_sum), _ptr(g[self.last_name]), _ptr(self.g["weight_output/kernel"]),
             _ptr(self.g["weight_output/bias"]), _ptr(self.ws), B * self.P * self.P, self.root, inv_count, st)
        # ---- decoder, stage L-2 .. 0
        for i in reversed(range(L - 1)):
            j, lvl = L + i, L - 2 - i
            c1, c2, up = a["c1_%d" % j], a["c2_%d" % j], a["up_%d" % i]
            h = up.shape[1]
            nf = c2.shape[3]
            dz2, dz1 = g["c2_%d" % j], g["c1_%d" % j]
            self._wgrad("conv_%d/conv2" % j, [(c1, h - 2)], dz2, h - 4)
            self._bwd_data("conv_%d/conv2" % j, dz2, dz1, h - 2, relu_src=c1)
            srcs = [(a["c2_%d" % lvl], h)] + ([(a["d2_%d" % lvl], h)] if self.dilated else []) + [(up, h)]
            self._wgrad("conv_%d/conv1" % j, srcs, dz1, h - 2)
            self._bwd_data("conv_%d/conv1" % j, dz1, g["skip_%d" % i], h, src_index=0)
            if self.dilated:
                self._bwd_data("conv_%d/conv1" % j, dz1, g["skipd_%d" % i], h, src_index=1)
            dup = g["up_%d" % i]
            self._bwd_data("conv_%d/conv1" % j, dz1, dup, h, src_index=2 if self.dilated else 1)
            # transposed conv
            upin = a["c2_%d" % (j - 1)] if i > 0 else a["c2_%d" % (L - 1)]
            gin = g["c2_%d" % (j - 1)] if i > 0 else g["c2_%d" % (L - 1)]
            if keep < 1.0:  # the transposed conv read the dropped tensor; (dropped > 0) = ReLU mask AND keep mask
                upin = a["drop_%d" % i]
            hh = h // 2
            self._wgradT(i, upin, dup, hh, nf)
            call("rsu_convT2x2_bwd_data", _ptr(dup), _ptr(self.pk["up_conv_%d/kernel" % i, "bwd"]), _ptr(gin), _ptr(upin), float(np.float32(1.0) / np.float32(keep)), B, hh, hh,
                 upin.shape[3], nf, self._ncu, st)
            self._grads_ready("up_conv_%d/kernel" % i)  # up_conv_i, conv_{L+i} and everything created later are final
        # ---- encoder, level L-1 .. 0
        for i in reversed(range(L)):
            c1, c2 = a["c1_%d" % i], a["c2_%d" % i]
            h = self.level_in[i]
            nf = c2.shape[3]
            dz2, dz1 = g["c2_%d" % i], g["c1_%d" % i]
            if i < L - 1:
                dec = L - 2 - i
                hs = a["up_%d" % dec].shape[1]
                call("rsu_pool_skip_relu_bwd_code", _ptr(c2), _ptr(self.pool_code.get(i)), _ptr(g["pool_%d" % i]), _ptr(g["skip_%d" % dec]), _ptr(dz2), B, h - 4, h - 4, nf, hs, hs,
                     keep, self.dropout_key(i + 1), st)
            self._wgrad("conv_%d/conv2" % i, [(c1, h - 2)], dz2, h - 4)
            self._bwd_data("conv_%d/conv2" % i, dz2, dz1, h - 2, relu_src=c1)
            if i > 0:
                pin = a["pool_%d" % (i - 1)]
                self._wgrad("conv_%d/conv1" % i, [(pin, h)], dz1, h - 2)
                self._bwd_data("conv_%d/conv1" % i, dz1, g["pool_%d" % (i - 1)], h)
            else:
                # the last group: nothing is left to run beside it on the main stream (unless the dilated twin of level 0 follows), so
                # it plans for the whole chip; the level-0 conv1 gradient below is a launch of its own
                self._flush_wgrads(alone=not (self.dilated and L > 1))
                if self.wstreams and self._split is not None and not (self.dilated and L > 1) and os.environ.get("RSU_TAIL_MAIN", "1") != "0":
                    # the pass's last launch needs the backward-data kernel that has just gone out on THIS stream and nothing runs beside
                    # it: on the main stream it follows back to back -- on the side stream it cost a fork and a join (~24 us of latency in
                    # the timeline). The side stream's last reduction must be done before the workspace is re-used: joined first.
                    self._join_side()
                    call("rsu_conv_first_bwd_weight", _ptr(self.in16), _ptr(dz1), _ptr(self.g["conv_0/conv1/kernel"]), _ptr(self.gfirst[0]),
                         _ptr(self.g["conv_0/conv1/bias"]), _ptr(self.ws_side[0]), B, h, h, nf, 1, self._split[0], st)
                else:
                    with UNet._Side(self, alone=not (self.dilated and L > 1)) as side:
                        call("rsu_conv_first_bwd_weight", _ptr(self.in16), _ptr(dz1), _ptr(self.g["conv_0/conv1/kernel"]), _ptr(self.gfirst[0]),
                             _ptr(self.g["conv_0/conv1/bias"]), _ptr(side.ws), B, h, h, nf, 1, self._ncu, self._stream())
            if self.dilated and i < L - 1:
                d1, d2 = a["d1_%d" % i], a["d2_%d" % i]
                dzd2, dzd1 = g["d2_%d" % i], g["d1_%d" % i]
                dec = L - 2 - i
                hs = a["up_%d" % dec].shape[1]
                call("rsu_pool_skip_relu_bwd", _ptr(d2), None, _ptr(g["skipd_%d" % dec]), _ptr(dzd2), B, h - 8, h - 8, nf, hs, hs, 1.0, 0, st)
                self._wgrad("conv_dilut_%d/atrous_conv2" % i, [(d1, h - 4)], dzd2, h - 8, dil=2)
                self._bwd_data("conv_dilut_%d/atrous_conv2" % i, dzd2, dzd1, h - 4, relu_src=d1, dil=2)
                if i > 0:
                    pin = a["pool_%d" % (i - 1)]
                    self._wgrad("conv_dilut_%d/atrous_conv1" % i, [(pin, h)], dzd1, h - 4, dil=2)
                    self._bwd_data("conv_dilut_%d/atrous_conv1" % i, dzd1, g["pool_%d" % (i - 1)], h, accumulate=1, dil=2)
                else:
                    with UNet._Side(self, alone=True) as side:
                        call("rsu_conv_first_bwd_weight", _ptr(self.in16), _ptr(dzd1), _ptr(self.g["conv_dilut_0/atrous_conv1/kernel"]),
                             _ptr(self.gfirst[1]), _ptr(self.g["conv_dilut_0/atrous_conv1/bias"]), _ptr(side.ws), B, h, h, nf, 2, self._ncu,
                             self._stream())
            if i > 0:
                # first LIVE variable of the level in creation order: the dilated pair of level L-1 is dead and sits behind n_live
                # (marking it ready launched nothing, and the largest block waited for level L-2)
                first_name = ("conv_dilut_%d/atrous_conv1/kernel" if (self.dilated and i < L - 1) else "conv_%d/conv1/kernel") % i
                self._grads_ready(first_name)
        self._flush_wgrads(alone=True)
        self._join_side()

    def tune(self, training=None, keep=1.0):
        """The explicit tile-shape tuning pass (rsu.h rsu_set_autotune): ONE untimed forward (+ backward) over random data with the
        library in RSU_TUNE_MEASURE mode -- every conv geometry of this network at the CU shares its launches use is timed once on an
        idle device -- then back to RSU_TUNE_LOOKUP, in which the launch entry points never measure nor synchronise. Weights, Momentum
        slots and the step counter are untouched; x / labels are restored. All shapes give the same bits: this only moves time. Under
        data parallelism call it before the first collective is in flight (and again after changing backward_cu_budget).
        `keep` is the dropout keep probability of the steps that follow: with keep < 1 the encoder's conv2 launches take the unfused
        conv + max-pool pair (rsu.h rsu_conv2d_fwd_pool folds the pool only at keep == 1), whose tuning keys differ from the fused
        launch's -- the pass must measure the keys the steps will look up."""
        lib = _lib.lib()
        training = self.training if training is None else (training and self.training)
        keep = float(keep)
        self._tuned.add((bool(training), self.backward_cu_budget, keep < 1.0))
        if lib.rsu_get_autotune() == _lib.TUNE_OFF or os.environ.get("RSU_AUTOTUNE", "1") == "0":
            return
        x0, l0 = self.x.clone(), self.labels.clone()
        g = torch.Generator(device="cpu").manual_seed(12345)
        self.x.copy_(torch.rand(tuple(self.x.shape), generator=g))
        self.labels.copy_((torch.rand(tuple(self.labels.shape), generator=g) < 0.2).to(torch.int64))
        on_grads, self.on_grads = self.on_grads, None   # no gradient exchange from inside the tuning pass
        prof, self.prof = self.prof, None
        call("rsu_set_autotune", _lib.TUNE_MEASURE)
        try:
            self.forward_device(keep=keep)
            if training:
                self.backward_device(1.0 / (self.B * self.P * self.P))
            if self.device.type == "cuda":
                torch.cuda.synchronize(self.device)
        finally:
            call("rsu_set_autotune", _lib.TUNE_LOOKUP)
            self.on_grads, self.prof = on_grads, prof
            self.x.copy_(x0)
            self.labels.copy_(l0)

    def ensure_tuned(self, training=None, keep=1.0):
        """tune() once per (forward-only / training, backward CU budget, dropout on / off): what the hosts call in front of their
        first step"""
        training = self.training if training is None else (training and self.training)
        drop = float(keep) < 1.0
        if (bool(training), self.backward_cu_budget, drop) not in self._tuned and ((True, self.backward_cu_budget, drop) not in self._tuned):
            self.tune(training, keep=keep)

    # ------------------------------------------------------------------ optimizer
    def learning_rate(self, lr0):
        """tf.train.exponential_decay(lr, global_step, 1000, 0.95, staircase=True) (tf_aerial_images.py:116-117), float32"""
        return float(np.float32(lr0) * np.float32(0.95) ** np.float32(self.global_step // 1000))

    def _build_update_table(self, pkset=0, rest_only=False):
        """the job table of rsu_update_table_run: one entry per conv / transposed-conv kernel (with its packed copies; the backward-data
        packs of set `pkset`), the variables between them (biases, colour adjust, the 1x1 head) as plain Momentum ranges; covers [0, n_live)
        of the flat buffers once. rest_only: without the kernels whose step rode on their weight-gradient launches (_fused_names)."""
        lib = _lib.lib()
        if not hasattr(self, "pk"):
            self.repack()
        skip = set(self._fused_names()) if rest_only else set()
        entries, plain_lo = [], None   # ("plain", lo, hi) / ("tensor", name, kind)
        live = sorted(((o, c, n, sh) for n, (o, c, sh) in self._slices.items() if o < self.n_live), key=lambda t: t[0])
        for o, c, n, sh in live:
            packed = n.endswith("kernel") and (n.startswith("up_conv") or (len(sh) == 4 and sh[0] == 3))
            if not packed:
                plain_lo = o if plain_lo is None else plain_lo
                continue
            if plain_lo is not None:
                entries.append(("plain", plain_lo, o))
                plain_lo = None
            if n not in skip:
                entries.append(("tensor", n, sh))
        if plain_lo is not None:
            entries.append(("plain", plain_lo, self.n_live))
        esz = lib.rsu_update_table_entry_bytes()
        host = ctypes.create_string_buffer(esz * len(entries))
        keep = []
        for idx, e in enumerate(entries):
            if e[0] == "plain":
                lo, hi = e[1], e[2]
                rc = lib.rsu_update_table_add_plain(host, idx, _ptr(self.flat_w[lo:hi]), _ptr(self.flat_acc[lo:hi]), _ptr(self.flat_g[lo:hi]), hi - lo)
            else:
                n, sh = e[1], e[2]
                w, a, g = _ptr(self.w[n]), _ptr(self.acc[n]), _ptr(self.g[n])
                bw = None
                if n.startswith("up_conv"):
                    cout, cin = sh[2], sh[3]
                    bw = (ctypes.c_void_p * 1)(self.pk[n, "bwd"].data_ptr()) if self.training else None
                    rc = lib.rsu_update_table_add(host, idx, 2, w, a, g, _ptr(self.pk[n, "fwd"]), bw, cin, cout, None, 0)
                elif sh[2] == NUM_CHANNELS:
                    rc = lib.rsu_update_table_add(host, idx, 4, w, a, g, _ptr(self.pk[n, "fwd"]), None, 3, sh[3], None, 0)
                else:
                    segs = self._conv_sources_c(n, sh)
                    packs = [self.pk[n, "bwd", si] if pkset == 0 else self._bwd_pack_set1(n, si) for si in range(len(segs))]
                    bw = (ctypes.c_void_p * len(segs))(*[t.data_ptr() for t in packs]) if self.training else None
                    rc = lib.rsu_update_table_add(host, idx, 0, w, a, g, _ptr(self.pk[n, "fwd"]), bw, sh[2], sh[3], (ctypes.c_int * len(segs))(*segs), len(segs))
                keep.append(bw)
            if rc != 1:
                raise _lib.RsuError("rsu_update_table_add(%s) failed: %d" % (e[1], rc))
        nb = ctypes.c_int(0)
        _lib.check(lib.rsu_update_table_finish(host, len(entries), ctypes.byref(nb)), "rsu_update_table_finish")
        tab = (torch.frombuffer(bytearray(host.raw), dtype=torch.uint8).to(self.device), len(entries), nb.value)
        self._update_tables[pkset, rest_only] = tab
        if pkset == 0 and not rest_only:
            self._update_table = tab   # (the name tools and probes know)
        return tab

    def apply_momentum(self, lr0, momentum, gscale=1.0):
        """MomentumOptimizer step on every live variable (tf_aerial_images.py:120-121) and the re-pack of the bf16 MFMA copies, in one
        pass over the parameters (rsu_update_table_run; RSU_FUSED_UPDATE=0: rsu_momentum_step, then the batched re-pack -- same bits).
        Behind backward_device(update=(lr0, momentum)) the 3x3 conv kernels have been stepped already: only the other variables are."""
        pending, self._fused_pending = self._fused_pending, None
        if pending is not None:
            if pending != (float(lr0), float(momentum)) or gscale != 1.0:
                raise _lib.RsuError("apply_momentum(%r, %r, gscale=%r) behind backward_device(update=%r): the step already applied to the conv kernels "
                                    "used other hyper-parameters" % (lr0, momentum, gscale, pending))
            tab = self._update_tables.get((0, True)) or self._build_update_table(0, True)
            call("rsu_update_table_run", _ptr(tab[0]), tab[1], tab[2], self.learning_rate(lr0), momentum, gscale, self._stream())
            self._pkset = 1 - self._pkset   # the fused updates wrote the other set of backward-data packs: the next step reads it
            self.global_step += 1
            return
        if os.environ.get("RSU_FUSED_UPDATE", "1") == "0":
            call("rsu_momentum_step", _ptr(self.flat_w), _ptr(self.flat_acc), _ptr(self.flat_g), self.learning_rate(lr0), momentum, gscale,
                 self.n_live, self._stream())
            self.global_step += 1
            self.repack()
            return
        tab = self._update_tables.get((self._pkset, False)) or self._build_update_table(self._pkset, False)
        call("rsu_update_table_run", _ptr(tab[0]), tab[1], tab[2], self.learning_rate(lr0), momentum, gscale, self._stream())
        self.global_step += 1


_DEFAULT_MODELS = {}


def forward(X, num_layers, root_size, dilated_layers, dropout_keep=None, params=None):
    """Drop-in for the reference's unet.forward (unet.py:12): X [B,S,S,3] float32 in [0,1] -> logits [B,P,P,2].

    The reference builds a TF graph and creates its variables on first use; here the variables live in a cached UNet
    keyed by the static shapes (pass `params` -- a dict of TF-named numpy arrays -- to set them). dropout_keep = None or
    1.0: no dropout (unet.py:29: `if dropout_keep is not None`); < 1: tf.nn.dropout at the 2L-1 sites with counter-based
    masks."""
    keep = 1.0 if dropout_keep is None else float(dropout_keep)
    X = torch.as_tensor(X)
    B, S = X.shape[0], X.shape[1]
    assert X.shape[2] == S and X.shape[3] == NUM_CHANNELS
    P = S - 12 * 2 ** (num_layers - 1) + 8
    assert input_size_needed(P, num_layers) == S, "input size {} is not a valid U-Net input for {} layers".format(S, num_layers)
    key = (num_layers, root_size, bool(dilated_layers), B, P)
    m = _DEFAULT_MODELS.get(key)
    if m is None:
        m = _DEFAULT_MODELS[key] = UNet(num_layers, root_size, dilated_layers, B, P, params=params, training=False)
    elif params is not None:
        m.load_state_dict(params)
    m.x.copy_(X.to(m.device, torch.float32))
    m.forward_device(want_logits=True, keep=keep)
    return m.logits
