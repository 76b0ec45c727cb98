"""-m gpu: the data-parallel training step on the real HIP path. Two processes share cuda:0 and exchange gradients over gloo
(RCCL needs one GPU per rank; the stream logic of dist.GradBucketer -- communication stream waiting on the backward-data and
weight-gradient streams, optimizer step after the reduced gradients -- is the same): after two steps on the two halves of a
batch the weights must equal those of one process stepping on the whole batch (tf_aerial_images.py:108 reduce_mean over the
global batch), up to fp32 summation order."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

L, ROOT, P, B = 3, 16, 20, 4


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _data():
    from oracle import unet_oracle as U
    S = U.input_size_needed(P, L)
    rng = np.random.RandomState(12)
    X = rng.rand(2, B, S, S, 3).astype(np.float32)
    labels = (rng.rand(2, B, P, P) < 0.3).astype(np.float64)
    params = U.init_params(L, ROOT, True, seed=13, bias_scale=0.05)
    return X, labels, params


def _images():
    return np.random.RandomState(14).rand(2, 44, 44, 3).astype(np.float32)


def _steps(model, X, labels, sl):
    for step in range(2):
        model.train_step(X[step][sl], labels[step][sl])
    torch.cuda.synchronize()
    sd = model.net.state_dict()
    return {k: v for k, v in sd.items() if not k.endswith("/Momentum") and k != "global_step"}


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_RANK="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from road_segmentation_unet_amd.model import ConvolutionalModel, Options
        X, labels, params = _data()
        m = ConvolutionalModel(Options(num_layers=L, root_size=ROOT, patch_size=P, batch_size=B, dilated_layers=True, dropout=1.0, lr=0.05,
                                       seed=3, stride=12, ensemble_prediction=True), device="cuda:0", params=params)
        assert m.world == world and m.local_batch == B // world
        per = B // world
        masks0 = m.predict(_images())          # (before training: weights identical to the single-process model's)
        w = _steps(m, X, labels, slice(rank * per, (rank + 1) * per))
        q.put((rank, w, masks0))
    finally:
        dist.destroy_process_group()


def test_two_rank_step_equals_single_process_step():
    from road_segmentation_unet_amd.model import ConvolutionalModel, Options
    X, labels, params = _data()
    single = ConvolutionalModel(Options(num_layers=L, root_size=ROOT, patch_size=P, batch_size=B, dilated_layers=True, dropout=1.0, lr=0.05,
                                        seed=3, stride=12, ensemble_prediction=True), device="cuda:0", params=params)
    ref_masks = single.predict(_images())
    ref = _steps(single, X, labels, slice(0, B))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(2)]
    got = {r[0]: r[1] for r in res}
    for r in res:   # inference: the phase classes of the shared-window sliding window are dealt over the ranks, the overlap accumulator all-reduced
        np.testing.assert_allclose(r[2], ref_masks, rtol=0, atol=1e-6)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for n in ref:
        np.testing.assert_array_equal(got[0][n], got[1][n], err_msg="ranks diverged: " + n)   # replicas stay identical
        upd = np.abs(ref[n] - params[n]).max()
        if upd == 0:
            continue
        # the two-rank gradient sums two 2-patch partial sums, the single process one 4-patch sum: bf16 activations identical,
        # only the fp32 reduction order differs
        assert np.abs(got[0][n] - ref[n]).max() <= 2e-2 * upd + 1e-7, (n, float(np.abs(got[0][n] - ref[n]).max()), float(upd))
