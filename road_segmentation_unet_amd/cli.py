"""Command-line surface of the reference's src/tf_aerial_images.py (flags :15-46, main :382-466) over the HIP path.

    python -m road_segmentation_unet_amd.cli --num_layers=5 --root_size=64 --patch_size=388 --dilated_layers ...
    torchrun --nproc-per-node 8 -m road_segmentation_unet_amd.cli ...     (data-parallel: --batch_size is the global batch)

Boolean flags follow tf.app.flags: `--flag`, `--flag=true|false`, `--noflag`."""
import argparse
import os
import sys
import time

import numpy as np


def build_parser():
    from .model import EXTRA_FLAG_DEFS, FLAG_DEFS
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    for name, typ, default, helptext in FLAG_DEFS + EXTRA_FLAG_DEFS:
        if typ is bool:
            ap.add_argument("--" + name, nargs="?", const=True, default=default,
                            type=lambda s: s.lower() in ("1", "true", "t", "yes", "y"), help=helptext)
            ap.add_argument("--no" + name, dest=name, action="store_false", help=argparse.SUPPRESS)
        else:
            ap.add_argument("--" + name, type=typ, default=default, help=helptext)
    return ap


def parse_options(argv=None):
    from .model import Options
    ns = build_parser().parse_args(argv)
    return Options(**vars(ns))


def main(argv=None):
    import torch
    import torch.distributed as dist
    from . import hostio
    from .model import ConvolutionalModel
    from .pool import DevicePatchPool
    from .unet import input_size_needed
    opts = parse_options(argv)
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and not dist.is_initialized():
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group("nccl")
    rank = dist.get_rank() if dist.is_initialized() else 0
    model = ConvolutionalModel(opts)
    print("Running on device {}".format(model.net.device))

    if opts.restore_model:
        if opts.model_path is not None:
            model.restore(file=opts.model_path)
            print("Restore model: {}".format(opts.model_path))
        else:
            print("Restore date: {}".format(opts.restore_date))
            model.restore(date=opts.restore_date, epoch=opts.restore_epoch)

    if opts.num_epoch > 0:
        train_images, train_groundtruth = hostio.load_train_data(opts.train_data_dir)
        input_size = input_size_needed(opts.patch_size, opts.num_layers)
        offset = int((input_size - opts.patch_size) / 2)
        extended = hostio.expand_and_rotate(train_images, opts.rotation_angles, offset)
        gt_exp = hostio.expand_and_rotate(train_groundtruth, opts.rotation_angles, 0)
        if opts.device_patch_pool:
            # the reference's extract_patches arrays as an index over the rotated images, resident in HBM (pool.DevicePatchPool):
            # same patches, same order, no 30-GB float64 pool on the host and no upload per step
            patches = DevicePatchPool(extended, gt_exp, input_size, opts.patch_size, opts.stride, device=model.net.device,
                                      augment=opts.d4_augmentation, seed=opts.seed)
            labels_patches = None
            print("Train on {} patches of size {}x{}".format(patches.shape[0], patches.shape[1], patches.shape[2]))
            print("Train on {} groundtruth patches of size {}x{}".format(patches.shape[0], opts.patch_size, opts.patch_size))
        else:
            patches = hostio.extract_patches(extended, patch_size=input_size, predict_patch_size=opts.patch_size, stride=opts.stride)
            print("Train on {} patches of size {}x{}".format(patches.shape[0], patches.shape[1], patches.shape[2]))
            labels_patches = hostio.extract_patches(gt_exp, patch_size=opts.patch_size, stride=opts.stride)
            print("Train on {} groundtruth patches of size {}x{}".format(*labels_patches.shape[:3]))
        summary = model._ensure_summary()
        if summary is not None:
            summary.add_to_eval_patch_summary(train_groundtruth)
        for i in range(opts.num_epoch):
            print("==== Train epoch: {} ====".format(i))
            if summary is not None:
                summary.reset()  # tf.local_variables_initializer().run(): reset scores
            stats = model.train(patches, labels_patches, train_images, train_groundtruth)
            if rank == 0:
                print("\nepoch {} : {}".format(i, stats))
            model.save(i)

    if opts.eval_train:
        # tf_aerial_images.py:432-445: predict the training images and dump five picture sets next to each other (file names as the
        # reference writes them, its "eval_orror" included, so that its downstream scripts keep finding them)
        print("Evaluate Test")
        eval_images, eval_groundtruth = hostio.load_train_data(opts.train_data_dir)
        probabilities = model.predict_batchwise(eval_images, opts.pred_batch_size)
        if rank == 0:
            binary = ((probabilities > 0.5) * 1).squeeze(-1)
            dumps = (
                ("eval_binary_pred_{:03d}.png", binary, True),
                ("eval_probability_pred_{:03d}.png", probabilities, True),
                ("eval_overlays_pred_{:03d}.png", hostio.overlays(eval_images, probabilities, fade=0.5), False),
                ("eval_confusion_{:03d}.png", hostio.overlap_pred_true(binary, eval_groundtruth), False),
                ("eval_orror_{:03d}.png", hostio.overlapp_error(binary, eval_groundtruth), True),
            )
            for pattern, pictures, grey in dumps:
                hostio.save_all(pictures, opts.eval_data_dir, pattern, greyscale=grey)

    if opts.eval_data_dir and not opts.eval_train:
        print("Running inference on eval data {}".format(opts.eval_data_dir))
        eval_images = hostio.load(opts.eval_data_dir)
        start = time.time()
        masks = model.predict_batchwise(eval_images, opts.pred_batch_size)
        print("Prediction time:{} mins".format((time.time() - start) / 60))
        masks = model.quantize_mask(masks, patch_size=hostio.IMG_PATCH_SIZE, threshold=hostio.FOREGROUND_THRESHOLD)
        if rank == 0:
            save_dir = os.path.abspath(os.path.join(opts.save_path, model.experiment_name))
            overlays = hostio.overlays(eval_images, masks, fade=0.4)
            hostio.save_all(overlays, save_dir)
            hostio.save_submission_csv(masks, save_dir, hostio.IMG_PATCH_SIZE)
            model.save_as(save_dir + "-model.chkpt")  # the model used for the submission
    if dist.is_initialized():
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
