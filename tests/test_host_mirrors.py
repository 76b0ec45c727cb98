"""CPU-side checks of the host mirrors of the reference interface: the 30 flags (names, types, defaults recorded from
the reference's own DEFINE_* calls), Options, and the off-path host data functions against the reference-generated goldens."""
import json
import os

import numpy as np
import pytest

from road_segmentation_unet_amd import hostio
from road_segmentation_unet_amd.cli import parse_options
from road_segmentation_unet_amd.model import FLAG_DEFS, Options, pixel_f1


def test_flags_match_reference(golden):
    ref = json.loads(str(golden["flags_json"]))
    assert len(ref) == 30 == len(FLAG_DEFS)
    kinds = {"integer": int, "boolean": bool, "float": float, "string": str}
    for (name, typ, default, _), (rname, rkind, rdefault) in zip(FLAG_DEFS, ref):
        assert name == rname
        assert typ is kinds[rkind], name
        if isinstance(rdefault, str) and rdefault.startswith("<abs>"):
            assert os.path.isabs(default) and default.endswith(os.path.basename(rdefault)), name  # os.path.abspath("./...")
        else:
            assert default == rdefault, name


def test_options_and_cli_parsing():
    o = Options()
    assert (o.batch_size, o.dropout, o.lr, o.momentum, o.num_layers, o.patch_size, o.root_size, o.stride, o.seed) == \
           (25, 0.8, 0.01, 0.9, 5, 128, 64, 16, 2017)
    assert o.rotation_angles is None and o.dilated_layers is False
    o = parse_options(["--num_layers=6", "--root_size", "64", "--patch_size=388", "--dilated_layers", "--rotation_angles=15,30,45",
                       "--noensemble_prediction", "--restore_model=true", "--dropout=1.0"])
    assert o.num_layers == 6 and o.patch_size == 388 and o.dilated_layers is True and o.rotation_angles == [15, 30, 45]
    assert o.ensemble_prediction is False and o.restore_model is True and o.dropout == 1.0
    with pytest.raises(AttributeError):
        Options(no_such_flag=1)


def test_hostio_against_reference_goldens(golden):
    np.testing.assert_array_equal(hostio.mirror_border(golden["g1_in4"], 3), golden["g1_out4_n3"])
    np.testing.assert_array_equal(hostio.extract_patches(golden["g2_in4"], 12, stride=4, predict_patch_size=4), golden["g2_out4_p12_s4_pp4"])
    np.testing.assert_array_equal(hostio.extract_patches(golden["g2_in3"], 6, stride=3), golden["g2_out3_p6_s3"])
    np.testing.assert_array_equal(hostio.quantize_mask(golden["g5_mask_in"], 0.25, 16), golden["g5_quant"])
    np.testing.assert_array_equal(hostio.labels_for_patches(golden["g5_lab_in"]), golden["g5_lab_out"])
    assert hostio.submission_rows(golden["g5_quant2"], 16) == str(golden["g5_csv"]).splitlines()[1:]
    lab = golden["g9_labels_img1"]
    mask = np.kron(lab.T, np.ones((16, 16)))[None, :, :, None]
    assert hostio.submission_rows(mask, 16) == str(golden["g9_rows_img1"]).splitlines()


def test_expand_and_rotate_golden(golden):
    """generated with the scipy of this image (the reference pinned scipy 1.0.0): same library on both sides here"""
    np.testing.assert_array_equal(hostio.expand_and_rotate(golden["g7_in"], [0, 15, 45], 6), golden["g7_out_off6"])
    np.testing.assert_array_equal(hostio.expand_and_rotate(golden["g7_in3"], [0, 30], 0), golden["g7_out3_off0"])
    with pytest.raises(TypeError):
        hostio.expand_and_rotate(golden["g7_in3"], None, 0)  # the reference also fails when --rotation_angles is not given


def test_csv_writer(tmp_path, golden):
    fn = hostio.save_submission_csv(golden["g5_quant2"], str(tmp_path), 16)
    assert open(fn).read() == str(golden["g5_csv"])


def test_pixel_f1():
    t = np.zeros((1, 4, 4)); t[0, :2] = 1
    p = np.zeros((1, 4, 4)); p[0, :2, :2] = 0.9; p[0, 3, 3] = 0.8
    # tp=4, fp=1, fn=4 -> precision 0.8, recall 0.5
    assert abs(pixel_f1(p, t) - 2 / (1 / 0.5 + 1 / 0.8)) < 1e-12
    assert pixel_f1(np.zeros_like(t), t) == 0.0


def test_postprocessing_images_against_reference_goldens(golden):
    """overlays / overlap_pred_true / overlapp_error / img_float_to_uint8 (images.py:19-21,102-128,282-309) and the label patches
    the metrics are computed on (summary.py:134-139, the in-place resize quirk included)"""
    np.testing.assert_array_equal(hostio.img_float_to_uint8(golden["g10_masks"]), golden["g10_u8"])
    np.testing.assert_array_equal(hostio.overlays(golden["g10_imgs"], golden["g10_masks"]), golden["g10_overlays_f095"])
    np.testing.assert_array_equal(hostio.overlays(golden["g10_imgs"], golden["g10_masks"], fade=0.4), golden["g10_overlays_f04"])
    np.testing.assert_array_equal(hostio.overlap_pred_true(golden["g10_pred"], golden["g10_true"]), golden["g10_overlap"])
    np.testing.assert_array_equal(hostio.overlapp_error(golden["g10_pred"], golden["g10_true"]), golden["g10_error"])
    got = hostio.img_to_label_patches(golden["g10_lab_in"])
    np.testing.assert_array_equal(got, golden["g10_label_patches"])
    n = got.shape[0]
    assert got.reshape(-1)[n:].sum() == 0  # the quirk: every real label sits in the first n entries, zeros behind


def test_rsu_plan_matches_the_python_shape_table():
    """rsu_plan (host arithmetic only, no GPU) against unet.param_shapes / input_size_needed and the buffer walk of UNet"""
    import ctypes

    from road_segmentation_unet_amd import _lib, unet
    lib = _lib.lib()
    for L, root, P, dil in [(3, 16, 188, 0), (5, 64, 388, 0), (6, 64, 388, 1), (4, 32, 92, 1)]:
        n = ctypes.c_int(0)
        tot = _lib.RsuPlanTotals()
        assert lib.rsu_plan(L, root, P, dil, 4, None, 0, ctypes.byref(n), ctypes.byref(tot)) == 0
        rows = (_lib.RsuPlanRow * n.value)()
        assert lib.rsu_plan(L, root, P, dil, 4, rows, n.value, ctypes.byref(n), ctypes.byref(tot)) == 0
        assert tot.input_size == unet.input_size_needed(P, L)
        assert tot.num_params == sum(int(np.prod(s)) for _, s in unet.param_shapes(L, root, bool(dil)))
        convs = [r for r in rows if r.kind == 1]
        # every 3x3 kernel of the live graph appears once, with the reference's channel counts
        want = [(s[2], s[3]) for nme, s in unet.param_shapes(L, root, bool(dil))
                if nme.endswith("kernel") and len(s) == 4 and s[0] == 3 and not unet._is_dead(nme, L)]
        assert sorted((r.Cin, r.Cout) for r in convs) == sorted(want)
        assert rows[0].kind == 0 and rows[0].Hin == tot.input_size and rows[n.value - 1].kind == 4 and rows[n.value - 1].Hout == P
        ups = [r for r in rows if r.kind == 3]
        assert len(ups) == L - 1 and all(r.Hout == 2 * r.Hin and r.Cin == 2 * r.Cout for r in ups)
        dec = [r for r in convs if r.nsrc > 1]
        assert len(dec) == L - 1 and all(r.nsrc == (3 if dil else 2) for r in dec)
        small = (_lib.RsuPlanRow * 2)()
        assert lib.rsu_plan(L, root, P, dil, 4, small, 2, ctypes.byref(n), None) == -12  # RSU_ENOMEM, count still reported
    assert lib.rsu_plan(5, 64, 390, 0, 4, None, 0, ctypes.byref(n), None) == -22  # the reference's input_size assertion


def test_cu_shares_of_a_data_parallel_budget():
    """unet.cu_shares (round 6): a budget of 224 .. 255 CUs keeps the weight-gradient stream at its 128 CUs and gives the backward-data stream the rest;
    256, smaller budgets and every non-default RSU_SPLIT_CHIP are shared out in proportion, in steps of 8, at least 32 CUs per stream (pure host logic)"""
    from road_segmentation_unet_amd.unet import cu_shares
    assert cu_shares(256, [128, 128]) == [128, 128]
    assert cu_shares(240, [128, 128]) == [112, 128]
    assert cu_shares(224, [128, 128]) == [96, 128]
    assert cu_shares(208, [128, 128]) == [104, 104]
    assert cu_shares(192, [128, 128]) == [96, 96]
    assert cu_shares(240, [112, 128]) == [104, 120]          # an explicit setting is scaled, not overridden
    assert cu_shares(256, [128, 64, 64]) == [128, 64, 64]
    assert cu_shares(64, [128, 128]) == [32, 32]
    for full in (256, 240, 224, 208, 192, 128):
        p = cu_shares(full, [128, 128])
        assert sum(p) <= full and min(p) >= 32 and all(v % 8 == 0 for v in p)
