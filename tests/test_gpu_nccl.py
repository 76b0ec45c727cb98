"""-m gpu: the data-parallel step and the sharded sliding-window prediction over REAL RCCL (torch.distributed backend "nccl"), one
process per GPU as bench.py / the CLI launch them. Same assertions as tests/test_gpu_dp.py (which runs the same code over gloo with
both ranks on one GPU). The two-rank test needs >= 2 GPUs (skipped on the one-GPU test box; fires on a multi-GPU node); the
single-rank test runs everywhere: a one-rank RCCL communicator, through which the bucketer's overlapped, tail-first all-reduces
really go (stream / event / async-work semantics of the RCCL backend, which gloo cannot stand in for)."""
import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = [pytest.mark.gpu]
two_gpus = pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one GPU per rank")

from tests.test_gpu_dp import B, L, P, ROOT, _data, _free_port, _images, _steps  # noqa: E402


def _worker(rank, world, port, q):
    import os

    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_RANK=str(rank), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    try:
        from road_segmentation_unet_amd.model import ConvolutionalModel, Options
        X, labels, params = _data()
        m = ConvolutionalModel(Options(num_layers=L, root_size=ROOT, patch_size=P, batch_size=B, dilated_layers=True, dropout=1.0, lr=0.05,
                                       seed=3, stride=12, ensemble_prediction=True, logdir=None), device="cuda:%d" % rank, params=params)
        per = B // world
        masks0 = m.predict(_images())
        w = _steps(m, X, labels, slice(rank * per, (rank + 1) * per))
        q.put((rank, w, masks0, m.exchange_schedule))
    finally:
        dist.destroy_process_group()


def _worker_one_rank(port, q):
    import os

    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        from road_segmentation_unet_amd.dist import GradBucketer
        from road_segmentation_unet_amd.unet import UNet
        out = {}
        for mode in ("plain", "overlapped", "single"):
            net = UNet(3, 16, True, 2, 60, device="cuda:0", seed=5, training=True)
            bucketer = None
            if mode != "plain":
                bucketer = GradBucketer(net.flat_g, net.n_live, min_bucket_elems=1 << 12, overlap=(mode == "overlapped"))
                bucketer.world = 2   # force the collective calls: over a one-rank communicator a SUM all-reduce is the identity
                bucketer.extra_streams = list(net.wstreams)
                net.on_grads = bucketer.ready
            g = torch.Generator(device="cpu").manual_seed(1)
            for _ in range(4):
                net.x.copy_(torch.rand((2, net.S, net.S, 3), generator=g))
                net.labels.copy_((torch.rand((2, 60, 60), generator=g) < 0.3).to(torch.int64))
                net.forward_device()
                if bucketer is not None:
                    bucketer.reset()
                net.backward_device(1.0 / (2 * 60 * 60))
                if bucketer is not None:
                    bucketer.finish()
                net.apply_momentum(0.05, 0.9)
            torch.cuda.synchronize()
            out[mode] = net.flat_w.cpu().numpy().copy()
        q.put((out, ".".join(str(v) for v in torch.cuda.nccl.version())))
    finally:
        dist.destroy_process_group()


def test_one_rank_rccl_bucketer_runs_the_real_collectives():
    """RCCL itself on the one-GPU box: the bucketer's tail-first all-reduces (communication stream, producer events, async work
    handles) and its single all-reduce behind backward, through a one-rank RCCL communicator, leave the training trajectory
    bit-identical to a run without any exchange"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker_one_rank, args=(_free_port(), q))
    p.start()
    out, ver = q.get(timeout=600)
    p.join(120)
    assert p.exitcode == 0
    assert ver
    np.testing.assert_array_equal(out["plain"], out["overlapped"])
    np.testing.assert_array_equal(out["plain"], out["single"])


@two_gpus
def test_two_gpu_rccl_step_and_sharded_prediction():
    from road_segmentation_unet_amd.model import ConvolutionalModel, Options
    X, labels, params = _data()
    single = ConvolutionalModel(Options(num_layers=L, root_size=ROOT, patch_size=P, batch_size=B, dilated_layers=True, dropout=1.0, lr=0.05,
                                        seed=3, stride=12, ensemble_prediction=True, logdir=None), device="cuda:0", params=params)
    ref_masks = single.predict(_images())
    ref = _steps(single, X, labels, slice(0, B))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(2)]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    got = {r[0]: r[1] for r in res}
    for r in res:
        np.testing.assert_allclose(r[2], ref_masks, rtol=0, atol=1e-6)   # accumulators all-reduced over RCCL
    for n in ref:
        np.testing.assert_array_equal(got[0][n], got[1][n], err_msg="ranks diverged: " + n)
        upd = np.abs(ref[n] - params[n]).max()
        if upd:
            assert np.abs(got[0][n] - ref[n]).max() <= 2e-2 * upd + 1e-7, n
