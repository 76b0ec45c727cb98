"""-m gpu: the command line of src/tf_aerial_images.py end to end on the HIP path, with the reference's DEFAULT dropout (0.8):
PNG training set -> rotation/expansion -> patches -> one training epoch -> checkpoint -> ensemble prediction -> submission CSV."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_cli_train_predict_submission(tmp_path):
    from PIL import Image
    from road_segmentation_unet_amd.cli import main
    rng = np.random.RandomState(4)
    tr, ev = tmp_path / "train", tmp_path / "eval"
    (tr / "images").mkdir(parents=True)
    (tr / "groundtruth").mkdir(parents=True)
    ev.mkdir()
    H = 48
    for i in range(3):
        img = (rng.rand(H, H, 3) * 255).astype(np.uint8)
        gt = ((img[..., 0] > 127) * 255).astype(np.uint8)
        Image.fromarray(img).save(tr / "images" / ("satImage_%03d.png" % i))
        Image.fromarray(gt).save(tr / "groundtruth" / ("satImage_%03d.png" % i))
    for i in range(2):
        Image.fromarray((rng.rand(H, H, 3) * 255).astype(np.uint8)).save(ev / ("test_%d.png" % i))
    save = tmp_path / "runs"
    argv = ["--num_layers=2", "--root_size=16", "--patch_size=16", "--stride=16", "--batch_size=4", "--num_epoch=1", "--lr=0.05",
            "--train_data_dir=%s" % tr, "--eval_data_dir=%s" % ev, "--save_path=%s" % save, "--pred_batch_size=2",
            "--rotation_angles=0,90", "--seed=5"]  # (like the reference, the training path needs the angles: None is not iterable there either)
    assert main(argv) == 0   # dropout stays at the flag default 0.8
    runs = [d for d in os.listdir(save) if os.path.isdir(save / d)]
    assert len(runs) == 1
    assert any(f.endswith("-model.chkpt.npz") for f in os.listdir(save))   # tf_aerial_images.py:458: the model used for the submission
    files = os.listdir(save / runs[0])
    assert any(f.endswith(".npz") for f in files), files
    csvs = [f for f in files if f.endswith(".csv")]
    assert len(csvs) == 1
    lines = open(save / runs[0] / csvs[0]).read().strip().split("\n")
    assert lines[0] == "id,prediction" and len(lines) == 1 + 2 * (H // 16) ** 2
