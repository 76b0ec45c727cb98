"""Host-side mirror of the reference's model driver (/root/reference/src/tf_aerial_images.py:51-379):
`Options` (the 30 flags) and `ConvolutionalModel` with train / predict / predict_batchwise / save / restore -- same names,
argument meaning, batch/tiling conventions and quirks that matter for parity -- over the HIP path (unet.UNet).

Data parallelism (new; the reference is single-device): when torch.distributed is initialised, `batch_size` is the GLOBAL
minibatch, sharded contiguously over ranks; gradients are all-reduced (dist.GradBucketer); predict() shards tiles.
"""
import glob
import math
import os
from datetime import datetime

import numpy as np
import torch
import torch.distributed as dist

from . import hostio
from . import images as dimages
from ._lib import call
from .dist import GradBucketer, shard_indices, tune_overlap
from .pool import BatchUploader, DevicePatchPool, PatchPool
from .summary import Summary
from .unet import UNet, input_size_needed

# (name, type, default, help) -- tf_aerial_images.py:15-46, same order
FLAG_DEFS = [
    ("batch_size", int, 25, "Batch size of training instances"),
    ("dilated_layers", bool, False, "Add dilated CNN layers"),
    ("dropout", float, 0.8, "Probability to keep an input"),
    ("ensemble_prediction", bool, False, "Ensemble Prediction"),
    ("eval_data_dir", str, None, "Directory containing eval images"),
    ("eval_every", int, 500, "Number of steps between evaluations"),
    ("eval_train", bool, False, "Evaluate training data"),
    ("gpu", int, -1, "GPU to run the model on"),
    ("image_augmentation", bool, False, "Augment training set of images with transformations"),
    ("interactive", bool, False, "Spawn interactive Tensorflow session"),
    ("logdir", str, os.path.abspath("./logdir"), "Directory where to write logfiles"),
    ("lr", float, 0.01, "Initial learning rate"),
    ("model_path", str, None, "Restore exact model path"),
    ("momentum", float, 0.9, "Momentum"),
    ("num_epoch", int, 5, "Number of pass on the dataset during training"),
    ("num_eval_images", int, 4, "Number of images to predict for an evaluation"),
    ("num_gpu", int, 1, "Number of available GPUs to run the model on"),
    ("num_layers", int, 5, "Number of layers of the U-Net"),
    ("patch_size", int, 128, "Size of the prediction image"),
    ("pred_batch_size", int, 2, "Batch size of batchwise prediction"),
    ("restore_date", str, None, "Restore the model from specific date"),
    ("restore_epoch", int, None, "Restore the model from specific epoch"),
    ("restore_model", bool, False, "Restore the model from previous checkpoint"),
    ("root_size", int, 64, "Number of filters of the first U-Net layer"),
    ("rotation_angles", str, None, "Rotation angles"),
    ("save_path", str, os.path.abspath("./runs"), "Directory where to write checkpoints, overlays and submissions"),
    ("seed", int, 2017, "Random seed for reproducibility"),
    ("stride", int, 16, "Sliding delta for patches"),
    ("train_data_dir", str, os.path.abspath("./data/training"), "Directory containing training images/ groundtruth/"),
    ("train_score_every", int, 1000, "Compute training score after the given number of iterations"),
]


# options the reference does not have (kept apart from its 30 flags): where the training patches live and the D4 augmentation
EXTRA_FLAG_DEFS = [
    ("device_patch_pool", bool, True, "Keep the rotated training images in HBM and cut the patches of a batch on the GPU"),
    ("d4_augmentation", bool, False, "Stochastic flips / transpose / rot90 per training sample on the GPU (what the reference's "
                                     "--image_augmentation subgraph intended; that flag itself stays without effect, as in the reference)"),
]


class Options(object):
    """Options used by our model (tf_aerial_images.py:51-84). Construct with keyword overrides of the flag defaults;
    `rotation_angles` accepts the flag string "a,b,c" and is stored as a list of ints like the reference."""

    def __init__(self, **overrides):
        for name, _typ, default, _help in FLAG_DEFS + EXTRA_FLAG_DEFS:
            setattr(self, name, default)
        for k, v in overrides.items():
            if not hasattr(self, k):
                raise AttributeError("unknown option %r" % k)
            setattr(self, k, v)
        ra = self.rotation_angles
        if isinstance(ra, str):
            self.rotation_angles = None if not ra else [int(i) for i in ra.split(",")]


def pixel_f1(pred_masks, true_masks, threshold=0.5):
    """F1 = 2 / (1/recall + 1/precision) (summary.py:141-147) at pixel level on binarised masks."""
    p = np.asarray(pred_masks).reshape(-1) > threshold
    t = np.asarray(true_masks).reshape(-1) >= 0.5
    tp = float(np.logical_and(p, t).sum())
    if tp == 0:
        return 0.0
    recall, precision = tp / t.sum(), tp / p.sum()
    return 2.0 / (1.0 / recall + 1.0 / precision)


class ConvolutionalModel:
    def __init__(self, options, session=None, device=None, params=None):
        self._options = opts = options
        self._session = session  # kept for signature parity; unused
        np.random.seed(opts.seed)
        self.input_size = input_size_needed(opts.patch_size, opts.num_layers)
        self.experiment_name = datetime.now().strftime("%Y-%m-%dT%Hh%Mm%Ss")
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        assert opts.batch_size % self.world == 0, "global batch_size must divide evenly over ranks"
        self.local_batch = opts.batch_size // self.world
        if device is None:
            device = "cuda:%d" % (opts.gpu if opts.gpu >= 0 else int(os.environ.get("LOCAL_RANK", "0")))
        if torch.device(device).type == "cuda":
            torch.cuda.set_device(torch.device(device))  # the library works on the HIP current device (rsu.h "devices")
        # the reference's graph is static in (batch, patch): one UNet serves training and (zero-padded) prediction batches
        self.net = UNet(opts.num_layers, opts.root_size, opts.dilated_layers, self.local_batch, opts.patch_size, device=device,
                        params=params, seed=opts.seed, training=True)
        self.net.dropout_seed = int(opts.seed) + 7919 * self.rank  # independent masks on every rank's shard
        self._bucketer = None
        self._exchange_tuned = False
        self.exchange_schedule = None
        self._uploader = None
        # summaries (tf_aerial_images.py:126-131,158-163): scalars loss + learning_rate per step, streaming train / eval scores
        # (created by the first train() call, so that prediction-only models leave no log directory behind)
        self._summary = None
        self._pending_scalars = []
        if self.world > 1:
            self._bucketer = GradBucketer(self.net.flat_g, self.net.n_live)
            self._bucketer.extra_streams = list(self.net.wstreams)
            self.net.on_grads = self._bucketer.ready
            # identical initial weights on every rank (the reference has one copy; ranks must start from the same point)
            dist.broadcast(self.net.flat_w, 0)
            self.net.repack()

    # ------------------------------------------------------------------ training
    def train_step(self, patches, labels):
        """One session.run([train, loss, predictions]) (tf_aerial_images.py:241-244) on this rank's shard.
        patches [b,S,S,3] float, labels [b,P,P] in {0,1}; returns (global mean loss tensor, predictions [b,P,P] device tensor).
        (Synchronous upload: the train() loop stages its batches one step ahead instead, see pool.BatchUploader.)"""
        net = self.net
        net.x.copy_(torch.as_tensor(np.asarray(patches, dtype=np.float32)).to(net.device))
        net.labels.copy_(torch.as_tensor(np.asarray(labels)).to(net.device, torch.int64))
        return self._run_step()

    def _run_step(self):
        """forward + loss + backward + gradient exchange + Momentum on the batch held in net.x / net.labels"""
        opts, net = self._options, self.net
        if self._bucketer is not None and not self._exchange_tuned:
            # first step of a data-parallel run: time forward + backward + exchange (no optimizer step, so the trajectory is
            # untouched) under both schedules and keep the faster one (dist.tune_overlap)
            self._exchange_tuned = True
            if "RSU_DP_OVERLAP" not in os.environ:
                def probe():
                    net.forward_device(keep=float(opts.dropout))
                    self._bucketer.reset()
                    net.backward_device(1.0 / (opts.batch_size * opts.patch_size * opts.patch_size))
                    self._bucketer.finish()
                def set_budget(n):   # every candidate budget gets its own (untimed) tile-shape tuning pass before it is timed
                    net.backward_cu_budget = n
                    net.ensure_tuned(keep=float(opts.dropout))
                self.exchange_schedule = tune_overlap(self._bucketer, probe, trials=2, set_cu_budget=set_budget)
        net.ensure_tuned(keep=float(opts.dropout))   # the explicit tile-shape tuning pass (untimed, weights untouched): once, in front of the first step
        # feed_dict dropout_keep: opts.dropout (tf_aerial_images.py:237); the masks come from a counter-based hash of
        # (seed, rank, dropout site, global step, element) instead of TF's Philox stream
        net.forward_device(keep=float(opts.dropout))
        if self._bucketer is not None:
            self._bucketer.reset()
        # (single device: Momentum + re-pack of the conv kernels fused into their weight-gradient launches, rsu.h rsu_conv2d_bwd_weight_update)
        net.backward_device(1.0 / (opts.batch_size * opts.patch_size * opts.patch_size),
                            update=(opts.lr, opts.momentum) if self._bucketer is None else None)
        loss = net.loss_sum / (opts.batch_size * opts.patch_size * opts.patch_size)
        if self._bucketer is not None:
            self._bucketer.finish()
            dist.all_reduce(loss)
        net.apply_momentum(opts.lr, opts.momentum)
        return loss, net.prob

    def _ensure_summary(self):
        opts = self._options
        if self._summary is None and self.rank == 0 and getattr(opts, "logdir", None):
            self._summary = Summary(opts, self._session, os.path.join(opts.logdir, self.experiment_name), device=self.net.device)
            self._summary.initialize_eval_summary()
            self._summary.initialize_train_summary()
            self._summary.initialize_overlap_summary()
            self._summary.initialize_missclassification_summary()
            self.summary_op = self._summary.get_summary_op({"loss": None, "learning_rate": None})
        return self._summary

    def _flush_scalars(self):
        """the per-step scalars are kept as device tensors and written in batches: reading them back every step would serialise
        the host with the GPU (the reference's session.run does exactly that)"""
        if self._summary is not None:
            for step, loss_t, err_t, total, lr in self._pending_scalars:
                self._summary.add({"loss": float(loss_t), "learning_rate": lr}, global_step=step)
                self._summary.add_to_pixel_missclassification_summary(float(err_t), total, step)
        self._pending_scalars = []

    def train(self, patches, labels_patches, imgs, labels):
        """Train the model for one epoch (tf_aerial_images.py:212-269): binarise labels at 0.5, shuffle with np.random,
        `for offset in range(0, N - batch_size, batch_size)` (the final batch is dropped even when full).
        `patches` is the reference's [N,S,S,3] array (then `labels_patches` its [N,P,P] labels) or a pool.PatchPool /
        pool.DevicePatchPool holding the same patches as an index (then `labels_patches` is ignored)."""
        opts, net = self._options, self.net
        self._ensure_summary()
        pool = patches if isinstance(patches, PatchPool) else None
        if pool is None:
            labels_patches = (np.asarray(labels_patches) >= 0.5) * 1.
        if labels is not None:
            labels = (np.asarray(labels) >= 0.5) * 1.
        num_train_patches = patches.shape[0]
        indices = np.arange(0, num_train_patches)
        np.random.shuffle(indices)
        num_errors = torch.zeros((), dtype=torch.float64, device=net.device)
        total = 0
        last = None
        offsets = list(range(0, num_train_patches - opts.batch_size, opts.batch_size))
        device_pool = isinstance(pool, DevicePatchPool)
        if not device_pool and self._uploader is None and net.device.type == "cuda":
            self._uploader = BatchUploader(net)

        def host_batch(offset):
            idx = shard_indices(indices, offset, opts.batch_size, self.rank, self.world)
            return pool.gather(idx) if pool is not None else (patches[idx], labels_patches[idx])

        if offsets and not device_pool:
            self._uploader.stage(0, *host_batch(offsets[0]))
        for batch_i, offset in enumerate(offsets):
            if device_pool:
                pool.load_batch(shard_indices(indices, offset, opts.batch_size, self.rank, self.world), net.x, net.labels)
            else:
                self._uploader.commit(batch_i & 1)
            loss, predictions = self._run_step()
            if not device_pool and batch_i + 1 < len(offsets):
                self._uploader.stage((batch_i + 1) & 1, *host_batch(offsets[batch_i + 1]))  # rides beside the step just launched
            step = net.global_step
            if self.rank == 0:
                print("Batch {} Step {}".format(batch_i, step), end="\r")
            num_errors += (net.labels.to(torch.float64) - predictions.to(torch.float64)).abs().sum()  # soft error (:249)
            total += opts.batch_size
            last = loss
            if self._summary is not None:
                self._pending_scalars.append((step, loss.clone(), num_errors.clone(), total, net.learning_rate(opts.lr)))
                if len(self._pending_scalars) >= 64:
                    self._flush_scalars()
            # from time to time do full prediction on some images (tf_aerial_images.py:253-264)
            if step > 0 and step % opts.eval_every == 0 and imgs is not None:
                images_to_predict = np.asarray(imgs)[:opts.num_eval_images]
                masks = self.predict(images_to_predict)
                if self.rank == 0:
                    f1 = pixel_f1(masks, labels[:opts.num_eval_images])
                    print("\nstep {} loss {:.5f} pixel-F1 on {} eval images {:.4f}".format(step, float(loss), opts.num_eval_images, f1))
                    if self._summary is not None:
                        overlays = hostio.overlays(images_to_predict, masks)
                        pred_masks = ((masks > 0.5) * 1).squeeze(-1)
                        self._summary.add_to_eval_summary(masks, overlays, labels, step)
                        self._summary.add_to_overlap_summary(labels[:opts.num_eval_images], pred_masks, step)
            if step > 0 and step % opts.train_score_every == 0 and imgs is not None:
                train_masks = self.predict(np.asarray(imgs))  # tf_aerial_images.py:266-267
                if self._summary is not None:
                    self._summary.add_to_training_summary(train_masks, labels, step)
        if self._summary is not None:
            self._flush_scalars()
            self._summary.flush()
        self.last_epoch_stats = {"loss": None if last is None else float(last), "soft_errors": float(num_errors), "patches": total}
        return self.last_epoch_stats

    # ------------------------------------------------------------------ inference
    @torch.no_grad()
    def predict(self, imgs):
        """Run inference on `imgs` and return predicted masks (tf_aerial_images.py:271-328).
        imgs: [num_images, H, H, 3] in [0,1]; returns numpy [num_images, H, H, 1] road probabilities.
        Ensemble x6 -> mirror border -> tiles (x-outer order) -> batched forward -> overlap average -> inverse ensemble.
        Tiles are sharded contiguously over ranks; the accumulators are summed with one all-reduce."""
        opts, net = self._options, self.net
        dev = net.device
        imgs_t = torch.as_tensor(np.asarray(imgs)).to(dev, torch.float32)
        num_images = imgs_t.shape[0]
        if opts.ensemble_prediction:
            imgs_t = dimages.image_augmentation_ensemble(imgs_t).contiguous()
            num_images = imgs_t.shape[0]
        H, P, S, B = imgs_t.shape[1], opts.patch_size, self.input_size, self.local_batch
        assert (H - P) % opts.stride == 0, "Stride sliding should cover the whole image"
        pps = (H - P) // opts.stride + 1
        num_patches = num_images * pps * pps
        acc = dimages.OverlapAccumulator(num_images, H, P, opts.stride, device=dev)
        was_training = net.training
        net.training = False
        reduced = False
        if os.environ.get("RSU_PREDICT_SHARED", "1") == "1" and pps > 1:
            # every rank fills the tiles of its own phase classes (zeros elsewhere) and overlap-adds the lot locally: the hit counts are
            # then complete on every rank, and ONE all-reduce of the accumulator (H*H floats per image variant, 1.4 MB at 604 px --
            # not the 1.3 GB of tiles) completes the sums
            tiles = self._shared_window_tiles(imgs_t, pps)
            acc.add(tiles.view(-1, P, P), 0)
            if self.world > 1:
                dist.all_reduce(acc.acc)
            reduced = True
        else:
            per = -(-num_patches // self.world)
            lo, hi = min(self.rank * per, num_patches), min((self.rank + 1) * per, num_patches)
            net.ensure_tuned(training=False)
            for t0 in range(lo, hi, B):
                nb = min(B, hi - t0)
                if nb < B:
                    net.x.zero_()  # the reference pads the last batch with zero patches (tf_aerial_images.py:298-301)
                dimages.extract_mirrored_patches(imgs_t, S, P, opts.stride, t0=t0, ntiles=nb, out=net.x[:nb])
                net.forward_device()
                acc.add(net.prob[:nb], t0)
        net.training = was_training
        if self.world > 1 and not reduced:
            dist.all_reduce(acc.acc)
            dist.all_reduce(acc.hits)
        masks = acc.finish()
        if opts.ensemble_prediction:
            masks = dimages.invert_image_augmentation_ensemble(masks)
        return masks.cpu().numpy()

    def _shared_window_tiles(self, imgs_t, pps):
        """The sliding window of tf_aerial_images.py:288-320 without its redundancy. The network is fully convolutional with VALID
        convolutions; its L-1 pools tie the result to the input offset modulo 2^(L-1) only. Tiles whose offsets agree modulo that
        period (on both axes) are therefore sub-windows of ONE forward pass over the union of their input windows: with stride 12
        and L = 6 the 19 x 19 tiles of a 604-pixel image fall into 8 x 8 phase classes of up to 3 x 3 tiles spaced lcm(12, 32) = 96
        pixels apart -- 64 passes over <= 956-pixel windows instead of 361 passes over 764-pixel tiles (3.6x fewer FLOPs).
        Every output element sees exactly the arithmetic of the per-tile pass (the reduction order of an output element does not
        depend on the tile it lies in), so the tiles are bit-identical; they are assembled in the reference's tile order and
        averaged by the same overlap kernel. Phase classes are dealt round-robin to the ranks.
        Returns float32 [n, pps, pps, P, P] indexed [image][x index][y index] (this rank's classes; zeros elsewhere)."""
        opts, net = self._options, self.net
        dev = net.device
        n, H = imgs_t.shape[0], imgs_t.shape[1]
        P, S, L, stride = opts.patch_size, self.input_size, opts.num_layers, opts.stride
        off = (S - P) // 2
        period = 2 ** (L - 1)
        g = stride * period // math.gcd(stride, period)   # spacing of same-phase tiles
        # symmetric ("mirror_border", images.py:269-281) padding by index arithmetic; one extra window of zeros behind it
        c = torch.arange(-off, H + off, device=dev)
        idx = torch.where(c < 0, -c - 1, torch.where(c >= H, 2 * H - c - 1, c))
        padded = imgs_t[:, idx][:, :, idx].contiguous()           # [n, Hp, Hp, 3]
        Hp = padded.shape[1]
        smax = int(os.environ.get("RSU_PREDICT_MAX_WINDOW", "1100"))   # largest input window (memory: ~1.7 GB per image at L = 6)
        kmax = max(1, 1 + (smax - S) // g)
        # per axis: tile indices of each phase class, cut into runs of <= kmax tiles spaced g apart
        step = g // stride
        runs = []
        for first in range(min(step, pps)):
            cls = list(range(first, pps, step))
            runs += [cls[i:i + kmax] for i in range(0, len(cls), kmax)]
        tiles = torch.zeros((n, pps, pps, P, P), dtype=torch.float32, device=dev)   # [img][xi][yi] = the reference's tile order
        jobs = [(rx, ry) for rx in runs for ry in runs]
        bmax = max(1, int(os.environ.get("RSU_PREDICT_WINDOW_BATCH", "6")))   # ~2 GB of activations per window image at L = 6
        Bw = max(d for d in range(1, min(n, bmax) + 1) if n % d == 0)
        for ji, (rx, ry) in enumerate(jobs):
            if ji % self.world != self.rank:
                continue
            k = max(len(rx), len(ry))
            Pk = P + (k - 1) * g
            wn = self._window_net(Pk, Bw)
            wn.ensure_tuned(training=False)   # the window nets have geometries of their own: one untimed tuning pass each
            Sk = wn.S
            ox0, oy0 = rx[0] * stride, ry[0] * stride
            h, w = min(Sk, Hp - oy0), min(Sk, Hp - ox0)
            for b0 in range(0, n, Bw):
                if h < Sk or w < Sk:
                    wn.x.zero_()
                wn.x[:, :h, :w] = padded[b0:b0 + Bw, oy0:oy0 + h, ox0:ox0 + w]
                wn.forward_device()
                for a, xi in enumerate(rx):
                    for bb, yi in enumerate(ry):
                        tiles[b0:b0 + Bw, xi, yi] = wn.prob[:, bb * g:bb * g + P, a * g:a * g + P]
        return tiles

    def _window_net(self, Pk, Bw):
        """a forward-only network for a larger output window, sharing this model's current weights"""
        nets = self.__dict__.setdefault("_win_nets", {})
        opts = self._options
        wn = nets.get((Pk, Bw))
        if wn is None:
            wn = nets[Pk, Bw] = UNet(opts.num_layers, opts.root_size, opts.dilated_layers, Bw, Pk, device=self.net.device,
                                     params=None, seed=opts.seed, training=False)
            wn._weights_version = None
        ver = (self.net.global_step, getattr(self.net, "_load_count", 0))
        if wn._weights_version != ver:
            wn.flat_w.copy_(self.net.flat_w)
            wn.repack()
            wn._weights_version = ver
        return wn

    def predict_batchwise(self, imgs, pred_batch_size):
        """tf_aerial_images.py:330-341"""
        masks = []
        for i in range(int(np.ceil(imgs.shape[0] / pred_batch_size))):
            start = i * pred_batch_size
            masks.append(self.predict(imgs[start:start + pred_batch_size]))
        return np.concatenate(masks, axis=0) if len(masks) > 1 else masks[0]

    def quantize_mask(self, masks, threshold, patch_size):
        """images.quantize_mask (images.py:256-266) on the device: masks [n, H, H, 1] float -> same shape, every patch_size block
        overwritten with its label mean(mask >= 0.5) > threshold"""
        import ctypes
        a = np.asarray(masks)
        t = torch.as_tensor(np.ascontiguousarray(a[..., 0], dtype=np.float32)).to(self.net.device)
        st = ctypes.c_void_p(torch.cuda.current_stream(self.net.device).cuda_stream)
        call("rsu_quantize_mask", ctypes.c_void_p(t.data_ptr()), ctypes.c_void_p(t.data_ptr()), t.shape[0], t.shape[1], int(patch_size),
             float(threshold), st)
        return t.cpu().numpy()[..., None].astype(a.dtype)

    # ------------------------------------------------------------------ checkpoints
    def save_as(self, path):
        """tf_aerial_images.py:458: the saver writing to an explicit path (`path`.npz: every variable under its TF name and layout,
        its Momentum slot and global_step; '/' in a name is stored as '|' -- np.savez keys become file names)"""
        if self.rank == 0:
            os.makedirs(os.path.dirname(path), exist_ok=True)
            np.savez(path + ".npz", **{k.replace("/", "|"): v for k, v in self.net.state_dict().items()})
        return path

    def save(self, epoch=0):
        """tf_aerial_images.py:343-349: {save_path}/{experiment_name}/model-epoch-{epoch:03d}.chkpt(.npz)"""
        opts = self._options
        path = self.save_as(os.path.abspath(os.path.join(opts.save_path, self.experiment_name, 'model-epoch-{:03d}.chkpt'.format(epoch))))
        if self.rank == 0:
            print("Model saved in file: {}".format(path))
        return path

    def restore_from_tf_arrays(self, arrays):
        """Load a checkpoint of the REFERENCE: `arrays` maps TensorFlow variable names to numpy arrays, as written by
        tools/export_tf_checkpoint.py on a machine that has TensorFlow (tf.train.load_checkpoint(...).get_tensor(name) for every
        name; tf_aerial_images.py:171 saves all global variables). Names and layouts are the reference's own, so this is a pure
        rename: `<var>` -> weights, `<var>/Momentum` -> optimizer slots, `global_step` (tf_aerial_images.py:113) -> step counter.
        A ':0' suffix and a leading scope are tolerated; missing variables raise KeyError."""
        clean = {}
        for k, v in arrays.items():
            k = k[:-2] if k.endswith(":0") else k
            clean[k] = np.asarray(v)
        d = {}
        for n in self.net.names:
            hit = [k for k in clean if k == n or k.endswith("/" + n)]
            if not hit:
                raise KeyError("variable %r not found in the exported checkpoint" % n)
            d[n] = clean[hit[0]]
            if d[n].shape != tuple(self.net.w[n].shape):
                raise ValueError("variable %r: checkpoint shape %s, network %s" % (n, d[n].shape, tuple(self.net.w[n].shape)))
            mom = [k for k in clean if k == n + "/Momentum" or k.endswith("/" + n + "/Momentum")]
            if mom:
                d[n + "/Momentum"] = clean[mom[0]]
        gs = [k for k in clean if k == "global_step" or k.endswith("/global_step")]
        d["global_step"] = int(clean[gs[0]]) if gs else 0
        self.net.load_state_dict(d)

    def restore(self, date=None, epoch=None, file=None):
        """Restores model from saved checkpoint (tf_aerial_images.py:351-379): explicit file, else newest experiment
        directory under save_path (or `date`), newest epoch (or `epoch`). As in the reference (tf_aerial_images.py:360), with
        date=None EVERY directory under save_path takes part in the "newest by name" choice, experiment directory or not."""
        opts = self._options
        if file is not None:
            model_data_dir = file
        else:
            if date is None:
                dates = [d for d in glob.glob(os.path.join(opts.save_path, "*")) if os.path.isdir(d)]
                model_data_dir = sorted(dates)[-1]
            else:
                model_data_dir = os.path.abspath(os.path.join(opts.save_path, date))
            if epoch is None:
                model_data_dir = sorted(glob.glob(os.path.abspath(os.path.join(model_data_dir, 'model-epoch-*.chkpt.npz'))))[-1][:-4]
            else:
                model_data_dir = os.path.abspath(os.path.join(model_data_dir, 'model-epoch-{:03d}.chkpt'.format(epoch)))
        path = model_data_dir if model_data_dir.endswith(".npz") else model_data_dir + ".npz"
        with np.load(path) as z:
            self.net.load_state_dict({k.replace("|", "/"): z[k] for k in z.files})
        print("Model restored from from file: {}".format(model_data_dir))
