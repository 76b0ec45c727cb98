"""Training / evaluation metrics and the event log -- the host-side mirror of /root/reference/src/summary.py (SURVEY.md
section 8(f) row f4). Same class and method names, argument meaning and quirks; what TensorBoard's protobuf event files were
there is a plain JSON-lines log here (`events.jsonl`: one {"step", "tag", "value"} object per scalar) plus PNG files for the image
summaries. The label-patch reduction and the tf.metrics counters run on the GPU through the C ABI (rsu_labels_for_patches,
rsu_confusion_counts); nothing here touches the training step's timed path.

Quirks kept on purpose (they change the numbers the reference logs):
  * `img_to_label_patches` (summary.py:134-139) resizes the [n] label vector IN PLACE to [n, 16, 16]; numpy fills the new
    entries with zeros, so every label is followed by 255 zeros in both predictions and labels. Recall, precision and F1 do not
    see them (no positives), accuracy does: (correct + 255 n) / (256 n).
  * the metrics are tf.metrics.* STREAMING metrics ([1] = the update op, summary.py:141-147): their counters run on until
    `reset()` (the reference runs tf.local_variables_initializer() once per epoch, tf_aerial_images.py:421).
  * the misclassification rate divides a PIXEL error count by the number of PATCHES seen (tf_aerial_images.py:249-251)."""
import ctypes
import json
import os

import numpy as np
import torch

from . import hostio
from ._lib import call

IMG_PATCH_SIZE = hostio.IMG_PATCH_SIZE
FOREGROUND_THRESHOLD = hostio.FOREGROUND_THRESHOLD


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


class StreamingMetrics:
    """Running TP / FP / FN / TN of tf.metrics.accuracy, .recall, .precision (one set of local variables per summary in the
    reference; they share their updates here). Counters live on the device; `values()` reads them back."""

    def __init__(self, device):
        self.device = torch.device(device)
        self.counts = torch.zeros(4, dtype=torch.int64, device=self.device)

    def reset(self):
        self.counts.zero_()

    def update(self, predictions, labels, padded_zeros=0):
        """predictions, labels: int64 device tensors of equal size in {0, 1}; padded_zeros: matching zero entries that the
        reference's resized label arrays carry behind the real labels (true negatives)."""
        n = predictions.numel()
        assert labels.numel() == n
        st = ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        call("rsu_confusion_counts", _ptr(predictions), _ptr(labels), n, _ptr(self.counts), st)
        if padded_zeros:
            self.counts[3] += int(padded_zeros)
        return self.values()

    def values(self):
        """(accuracy, recall, precision, f1_score) with f1 = 2 / (1/recall + 1/precision) (summary.py:145); tf.metrics return 0
        where a denominator is 0, and 1/0 = inf makes the F1 0 there."""
        tp, fp, fn, tn = (float(v) for v in self.counts.tolist())
        total = tp + fp + fn + tn
        accuracy = (tp + tn) / total if total else 0.0
        recall = tp / (tp + fn) if tp + fn else 0.0
        precision = tp / (tp + fp) if tp + fp else 0.0
        f1 = 0.0 if recall == 0.0 or precision == 0.0 else 2.0 / (1.0 / recall + 1.0 / precision)
        return accuracy, recall, precision, f1


class Summary:
    """Handle summaries (summary.py:7-147). `session` is kept for signature parity and unused."""

    def __init__(self, options, session, summary_path, device="cuda:0"):
        self._options = options
        self._session = session
        self._path = summary_path
        self._device = torch.device(device)
        os.makedirs(summary_path, exist_ok=True)
        self._events = open(os.path.join(summary_path, "events.jsonl"), "a")
        self.summary_ops = []
        self._train_metrics = self._eval_metrics = None

    # ---- event log -----------------------------------------------------------------------------
    def flush(self):
        self._events.flush()

    def add(self, scalars, global_step=None):
        """summary.py:19-20 add(summary_str, global_step): here `scalars` is the {tag: value} dict that get_summary_op merged."""
        for tag, value in scalars.items():
            self._events.write(json.dumps({"step": None if global_step is None else int(global_step), "tag": tag, "value": float(value)}) + "\n")

    def get_summary_op(self, scalars):
        """summary.py:22-26: remembers the tags; returns them (the per-step values go through add())."""
        self.summary_ops += list(scalars.keys())
        return list(self.summary_ops)

    def _add_images(self, tag, imgs, global_step, max_outputs):
        from PIL import Image
        for i in range(min(len(imgs), max_outputs)):
            a = np.asarray(imgs[i])
            if a.ndim == 3 and a.shape[-1] == 1:
                a = a[..., 0]
            Image.fromarray(a).save(os.path.join(self._path, "step_{:07d}_{}_{}.png".format(int(global_step or 0), tag, i)))

    # ---- initialisers (summary.py:28-78) ---------------------------------------------------------
    def initialize_eval_summary(self):
        self._eval_metrics = StreamingMetrics(self._device)

    def initialize_overlap_summary(self):
        pass

    def initialize_train_summary(self):
        self._train_metrics = StreamingMetrics(self._device)

    def initialize_missclassification_summary(self):
        pass

    def reset(self):
        """tf.local_variables_initializer().run() (tf_aerial_images.py:421): zero the streaming counters"""
        for m in (self._train_metrics, self._eval_metrics):
            if m is not None:
                m.reset()

    # ---- summaries -------------------------------------------------------------------------------
    def add_to_overlap_summary(self, true_labels, predicted_labels, global_step):
        """summary.py:80-88"""
        overlapped = hostio.overlap_pred_true(np.asarray(predicted_labels), np.asarray(true_labels))
        self._add_images("groundtruth_vs_prediction", overlapped, global_step, self._options.num_eval_images)

    def add_to_eval_patch_summary(self, labels):
        """summary.py:90-99"""
        opts = self._options
        eval_labels = hostio.img_float_to_uint8(np.asarray(labels)[:opts.num_eval_images, :, :])
        self._add_images("eval_groundtruth", eval_labels, 0, eval_labels.shape[0])

    def add_to_pixel_missclassification_summary(self, num_errors, total, global_step):
        """summary.py:101-104"""
        self.add({"misclassification_rate": float(num_errors) / float(total)}, global_step)

    def add_to_eval_summary(self, masks, overlays, labels, global_step):
        """summary.py:106-121"""
        opts = self._options
        eval_pred = self.img_to_label_patches(masks)
        eval_true = self.img_to_label_patches(np.asarray(labels)[:opts.num_eval_images, :, :])
        acc, rec, prec, f1 = self._update(self._eval_metrics, eval_pred, eval_true)
        self._add_images("eval_masks", hostio.img_float_to_uint8(np.asarray(masks)), global_step, opts.num_eval_images)
        self._add_images("eval_images", overlays, global_step, opts.num_eval_images)
        self.add({"eval accuracy": acc, "eval recall": rec, "eval precision": prec, "eval f1_score": f1}, global_step)
        return acc, rec, prec, f1

    def add_to_training_summary(self, predictions, labels, global_step):
        """summary.py:123-132"""
        train_predictions = self.img_to_label_patches(predictions)
        train_labels = self.img_to_label_patches(labels)
        acc, rec, prec, f1 = self._update(self._train_metrics, train_predictions, train_labels)
        self.add({"train accuracy": acc, "train recall": rec, "train precision": prec, "train f1_score": f1}, global_step)
        return acc, rec, prec, f1

    def _update(self, metrics, pred, true):
        (p, pad_p), (t, pad_t) = pred, true
        assert pad_p == pad_t
        return metrics.update(p, t, padded_zeros=pad_p)

    def img_to_label_patches(self, img, patch_size=IMG_PATCH_SIZE):
        """summary.py:134-139 on the device: per 16x16 patch (x outer, y inner) label = mean > 0.25. Returns (int64 device
        tensor of the n real labels, number of zeros the reference's in-place resize to [n, 16, 16] puts behind them)."""
        a = np.asarray(img)
        if a.ndim == 4 and a.shape[-1] == 1:
            a = a[..., 0]
        assert a.ndim == 3 and a.shape[1] == a.shape[2], "Assume square images"
        nimg, S = a.shape[0], a.shape[1]
        assert S % patch_size == 0, "Stride sliding should cover the whole image"
        m = torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32)).to(self._device)
        nb = S // patch_size
        labels = torch.empty((nimg, nb, nb), dtype=torch.int64, device=self._device)
        st = ctypes.c_void_p(torch.cuda.current_stream(self._device).cuda_stream)
        call("rsu_labels_for_patches", _ptr(m), _ptr(labels), nimg, S, patch_size, float(FOREGROUND_THRESHOLD), st)
        n = nimg * nb * nb
        return labels.view(-1), n * (patch_size * patch_size - 1)

    def get_prediction_metrics(self, labels, predictions, metrics=None):
        """summary.py:141-147 as a function of int64 device tensors: updates the streaming counters and returns
        (accuracy, recall, precision, f1_score)"""
        metrics = metrics if metrics is not None else StreamingMetrics(self._device)
        return metrics.update(predictions, labels)
