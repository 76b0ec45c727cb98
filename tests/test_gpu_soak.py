"""-m gpu: run-to-run repeatability of the whole config-2 step. Every kernel has a fixed reduction order, so repeated passes
over the same inputs must agree bit for bit; a register hazard in the compiled code (tools/check_mfma_hazards.py has the
rules) shows up here as a handful of differing elements once in some tens of passes."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.gpu


def test_c2_passes_are_bit_identical():
    import soak_net
    # the epilogue-store hazard fixed in round 1 corrupted one pass in ~40; 200 passes would have caught it with p > 0.99
    assert soak_net.soak(200, bwd=True) == 0


def test_c2_passes_with_dropout_and_small_net_are_bit_identical():
    import soak_net
    assert soak_net.soak(60, bwd=True, keep=0.8) == 0
    assert soak_net.soak(150, bwd=True, L=3, root=32, P=100, B=2) == 0


def test_c3_dilated_passes_are_bit_identical():
    import soak_net
    assert soak_net.soak(80, bwd=True, L=6, B=1, dilated=True) == 0   # dilated twins, three-source decoder convs, accumulate


def test_short_training_runs_are_bit_identical():
    from road_segmentation_unet_amd.unet import UNet

    def run():
        m = UNet(5, 64, False, 4, 388, seed=2018, training=True)
        g = torch.Generator(device="cpu").manual_seed(7)
        for _ in range(40):
            m.x.copy_(torch.rand((4, m.S, m.S, 3), generator=g))
            m.labels.copy_((torch.rand((4, 388, 388), generator=g) < 0.2).to(torch.int64))
            m.forward_device(keep=0.9)
            m.backward_device(1.0 / (4 * 388 * 388))
            m.apply_momentum(0.01, 0.9)
        torch.cuda.synchronize()
        return m.flat_w.clone()

    a, b = run(), run()
    assert bool(torch.isfinite(a).all()) and torch.equal(a, b)


def test_two_stream_schedule_does_not_change_the_numbers(monkeypatch):
    """The weight-gradient launches run on side streams by default; the same launches on one stream must give the same bits (with
    every launch planned for the whole chip: RSU_SPLIT_CHIP=0, one launch per layer: RSU_WG_GROUP=0), and the same numbers to
    summation order with the chip shared out between the streams (fewer, longer partial sums per weight gradient) or with the weight
    gradients grouped into few launches (RSU_WG_GROUP: the default on one stream is ONE group behind the pass)."""
    from road_segmentation_unet_amd.unet import UNet

    def run(single_stream, split="0", group="0", budget=None):
        monkeypatch.setenv("RSU_SPLIT_CHIP", split)
        if group is None:
            monkeypatch.delenv("RSU_WG_GROUP", raising=False)
        else:
            monkeypatch.setenv("RSU_WG_GROUP", group)
        m = UNet(4, 32, True, 2, 204, seed=11, training=True)
        m.backward_cu_budget = budget
        if single_stream:
            m.wstream, m.wstreams = None, []
        g = torch.Generator(device="cpu").manual_seed(3)
        for _ in range(6):
            m.x.copy_(torch.rand((2, m.S, m.S, 3), generator=g))
            m.labels.copy_((torch.rand((2, 204, 204), generator=g) < 0.2).to(torch.int64))
            m.forward_device(keep=0.8)
            m.backward_device(1.0 / (2 * 204 * 204))
            m.apply_momentum(0.01, 0.9)
        torch.cuda.synchronize()
        return m.flat_w.clone()

    a, b = run(False), run(True)
    assert torch.equal(a, b)
    for split in ("128,128", "128,64,64"):
        c = run(False, split)
        assert float((a - c).abs().max()) <= 1e-5 * float(a.abs().max()), split
    for single, split, group in ((True, "0", None), (False, "128,128", "1"), (False, "128,128", "2,3"), (True, "0", "all")):
        c = run(single, split, group)
        assert float((a - c).abs().max()) <= 1e-5 * float(a.abs().max()), (single, split, group)
        assert torch.equal(c, run(single, split, group)), "a grouped schedule must repeat bit for bit"
    # the budgets a data-parallel run offers (dist.EXCHANGE_CANDIDATES; round 6: 240 = 112 + 128, 224 = 96 + 128, 208 = 104 + 104): the weight gradients
    # keep the summation order of their CU share, the backward-data launches give the same bits at any share
    for budget in (240, 224, 208):
        c = run(False, "128,128", "0", budget)
        assert float((a - c).abs().max()) <= 1e-5 * float(a.abs().max()), budget
        assert torch.equal(c, run(False, "128,128", "0", budget)), "a budgeted schedule must repeat bit for bit"


def test_measured_tile_shapes_do_not_change_the_numbers():
    """rsu_set_autotune: the tile shapes picked by measurement give the same bits as the cost model's choice."""
    from road_segmentation_unet_amd._lib import call, lib
    from road_segmentation_unet_amd.unet import UNet

    def run(tune):
        call("rsu_set_autotune", 1 if tune else 0)   # RSU_TUNE_LOOKUP / RSU_TUNE_OFF
        m = UNet(4, 32, False, 2, 204, seed=5, training=True)
        if tune:
            m.tune()   # the explicit tuning pass: the only place where shapes are measured
            assert lib().rsu_get_autotune() == 1
        g = torch.Generator(device="cpu").manual_seed(9)
        for _ in range(3):
            m.x.copy_(torch.rand((2, m.S, m.S, 3), generator=g))
            m.labels.copy_((torch.rand((2, 204, 204), generator=g) < 0.2).to(torch.int64))
            m.forward_device(keep=0.9)
            m.backward_device(1.0 / (2 * 204 * 204))
            m.apply_momentum(0.01, 0.9)
        torch.cuda.synchronize()
        return m.flat_w.clone()

    try:
        n0 = lib().rsu_autotune_entries()
        a = run(True)
        assert lib().rsu_autotune_entries() > n0
        b = run(False)
    finally:
        call("rsu_set_autotune", 1)
    assert torch.equal(a, b)


@pytest.mark.parametrize("L,root,dilated,P", [(4, 32, False, 204), (3, 16, True, 60), (3, 64, True, 28), (2, 16, False, 20)])
def test_fused_update_equals_momentum_then_repack(L, root, dilated, P, monkeypatch):
    """rsu_update_table_run (Momentum + both packed layouts from ONE read of the weights) against rsu_momentum_step followed by the
    batched re-pack: weights, Momentum slots and every packed buffer bit for bit, over several steps, for nets with 16-channel concat
    segments (padded K chunks), dilated twins, transposed convs and the 3-channel first conv"""
    from road_segmentation_unet_amd.unet import UNet

    def run(fused):
        monkeypatch.setenv("RSU_FUSED_UPDATE", "1" if fused else "0")
        m = UNet(L, root, dilated, 2, P, seed=17, training=True)
        g = torch.Generator(device="cpu").manual_seed(6)
        for _ in range(3):
            m.x.copy_(torch.rand((2, m.S, m.S, 3), generator=g))
            m.labels.copy_((torch.rand((2, P, P), generator=g) < 0.2).to(torch.int64))
            m.forward_device()
            m.backward_device(1.0 / (2 * P * P))
            m.apply_momentum(0.05, 0.9)
        torch.cuda.synchronize()
        return m.flat_w.clone(), m.flat_acc.clone(), {k: v.clone() for k, v in m.pk.items()}

    a, b = run(True), run(False)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    for k in a[2]:
        assert torch.equal(a[2][k].view(torch.int16), b[2][k].view(torch.int16)), k


@pytest.mark.parametrize("L,root,dilated,P,B", [(5, 64, False, 388, 4), (6, 64, True, 388, 1), (4, 32, False, 204, 2), (3, 16, True, 60, 2), (3, 64, True, 28, 2)])
def test_update_fused_into_the_weight_gradient_side_equals_the_plain_step(L, root, dilated, P, B, monkeypatch):
    """backward_device(update=(lr, mu)) + apply_momentum (rsu_conv2d_bwd_weight_update: the launch that sums a conv kernel's weight-gradient
    slabs applies Momentum and writes both packed layouts, the backward-data packs double-buffered) against the plain backward_device +
    apply_momentum: weights, Momentum slots, the forward packs and the backward-data packs the NEXT step reads, bit for bit over three
    steps -- on c2 and c3 at full size and on small nets with 16-channel concat segments, dilated twins and launches
    that need no slabs. The third step of the fused run also asks for the gradients (keep_grad) and must give the plain run's."""
    from road_segmentation_unet_amd.unet import UNet
    monkeypatch.setenv("RSU_FUSED_WGRAD", "1")   # (opt-in: the schedule is slower than the plain one on this hardware, unet.py backward_device)

    def run(fused):
        m = UNet(L, root, dilated, B, P, seed=17, training=True)
        m.tune()
        g = torch.Generator(device="cpu").manual_seed(6)
        for step in range(3):
            m.x.copy_(torch.rand((B, m.S, m.S, 3), generator=g))
            m.labels.copy_((torch.rand((B, P, P), generator=g) < 0.2).to(torch.int64))
            m.forward_device()
            if fused:
                m.backward_device(1.0 / (B * P * P), update=(0.05, 0.9), keep_grad=(step == 2))
                assert m._fused_pending is not None, "the fused schedule did not run"
            else:
                m.backward_device(1.0 / (B * P * P))
            m.apply_momentum(0.05, 0.9)
        torch.cuda.synchronize()
        packs = {k: v.clone() for k, v in m.pk.items() if k[1] == "fwd" or len(k) == 2}
        for k in m.pk:
            if len(k) == 3:
                packs[k] = m._bwd_pack(k[0], k[2]).clone()
        return m.flat_w.clone(), m.flat_acc.clone(), packs, m.flat_g.clone(), m.global_step

    a, b = run(True), run(False)
    assert a[4] == b[4] == 3
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    for k in b[2]:
        assert torch.equal(a[2][k].view(torch.int16), b[2][k].view(torch.int16)), k
    assert torch.equal(a[3], b[3])
