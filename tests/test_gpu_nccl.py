"""-m gpu, needs >= 2 GPUs (skipped on the one-GPU test box; fires on a multi-GPU node): the data-parallel step and the sharded
sliding-window prediction over REAL RCCL (torch.distributed backend "nccl"), one process per GPU as bench.py / the CLI launch
them. Same assertions as tests/test_gpu_dp.py (which runs the same code over gloo with both ranks on one GPU)."""
import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one GPU per rank")]

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
