"""Pins oracle/tiler_oracle.py against golden vectors generated from the reference's own images.py
(tests/golden/make_golden.py). Bit-exact: this is index/byte work plus float64 averaging."""
import numpy as np
import pytest

from oracle import tiler_oracle as T
from oracle import unet_oracle as U


def test_g1_mirror_border(golden):
    np.testing.assert_array_equal(T.mirror_border(golden["g1_in4"], 3), golden["g1_out4_n3"])
    np.testing.assert_array_equal(T.mirror_border(golden["g1_in3"], 2), golden["g1_out3_n2"])
    np.testing.assert_array_equal(T.mirror_border(golden["g1_in4"], 7), golden["g1_out4_n7"])


def test_g2_extract_patches(golden):
    out = T.extract_patches(golden["g2_probe_in"], 2, stride=2)
    np.testing.assert_array_equal(out, golden["g2_probe_out"])
    assert [int(p[0, 0]) for p in out[:4]] == [0, 12, 24, 2]  # x outer, y inner
    for key, kw in [("g2_out4_p8_s4", dict(patch_size=8, stride=4)),
                    ("g2_out4_p12_s4_pp4", dict(patch_size=12, stride=4, predict_patch_size=4))]:
        o = T.extract_patches(golden["g2_in4"], **kw)
        assert o.dtype == np.float64
        np.testing.assert_array_equal(o, golden[key])
    np.testing.assert_array_equal(T.extract_patches(golden["g2_in3"], 4), golden["g2_out3_p4"])
    np.testing.assert_array_equal(T.extract_patches(golden["g2_in3"], 6, stride=3), golden["g2_out3_p6_s3"])


def test_g2_asserts():
    with pytest.raises(AssertionError):
        T.extract_patches(np.zeros((1, 10, 10)), 4, stride=4)  # (10-4) % 4 != 0
    with pytest.raises(AssertionError):
        T.extract_patches(np.zeros((1, 10, 12)), 2)            # not square
    with pytest.raises(AssertionError):
        T.extract_patches(np.zeros((1, 12, 12)), 4, predict_patch_size=3)


def test_g3_images_from_patches(golden):
    pt = T.extract_patches(golden["g2_in4"], 8, stride=4).reshape(2, -1, 8, 8, 3)
    np.testing.assert_array_equal(T.images_from_patches(pt, stride=4), golden["g3_roundtrip"])
    np.testing.assert_array_equal(T.images_from_patches(golden["g3_in"], stride=3), golden["g3_out_s3"])
    np.testing.assert_array_equal(T.images_from_patches(golden["g3_in"]), golden["g3_out_nostride"])


def test_g4_ensemble(golden):
    np.testing.assert_array_equal(T.image_augmentation_ensemble(golden["g4_in"]), golden["g4_aug"])
    np.testing.assert_array_equal(T.invert_image_augmentation_ensemble(golden["g4_masks"]), golden["g4_inv"])
    rt = T.invert_image_augmentation_ensemble(T.image_augmentation_ensemble(golden["g4_in"]))
    np.testing.assert_array_equal(rt, golden["g4_roundtrip"])
    np.testing.assert_allclose(rt, golden["g4_in"], rtol=0, atol=1e-7)


def test_g5_postprocessing(golden):
    np.testing.assert_array_equal(T.quantize_mask(golden["g5_mask_in"], 0.25, 16), golden["g5_quant"])
    np.testing.assert_array_equal(T.quantize_mask(golden["g5_mask2_in"], 0.25, 16), golden["g5_quant2"])
    np.testing.assert_array_equal(T.labels_for_patches(golden["g5_lab_in"]), golden["g5_lab_out"])
    csv = str(golden["g5_csv"]).splitlines()
    assert csv[0] == "id,prediction"
    assert T.submission_rows(golden["g5_quant2"], 16) == csv[1:]


def test_g6_input_size_needed(golden):
    for L, P, S in golden["g6_table"]:
        if S < 0:
            with pytest.raises(AssertionError):
                U.input_size_needed(int(P), int(L))
        else:
            assert U.input_size_needed(int(P), int(L)) == S
            assert S == P + 12 * 2 ** (L - 1) - 8
    with pytest.raises(AssertionError) as ei:
        U.input_size_needed(128, 5)
    assert str(ei.value) == str(golden["g6_assert_msg"])
    assert U.input_size_needed(188, 3) == 228 and U.input_size_needed(388, 5) == 572 and U.input_size_needed(388, 6) == 764


def test_g8_predictions_to_patches(golden):
    np.testing.assert_array_equal(T.predictions_to_patches(golden["g8_in"], 4), golden["g8_out"])


def test_g9_real_submission_block(golden):
    """CSV rows of image 001 of a shipped submission <-> 38x38 label grid, through the oracle's row writer."""
    lab = golden["g9_labels_img1"]  # indexed [x // 16][y // 16]: ids are '{img}_{x}_{y}' (images.py:232-236)
    mask = np.kron(lab.T, np.ones((16, 16)))[None, :, :, None].astype(np.float64)
    assert T.submission_rows(mask, 16) == str(golden["g9_rows_img1"]).splitlines()
