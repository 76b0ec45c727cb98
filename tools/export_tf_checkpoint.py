#!/usr/bin/env python3
"""Run this where TensorFlow is installed (it is not in this repository's image): dumps every variable of a checkpoint written by
the reference (tf.train.Saver, tf_aerial_images.py:171,343-349) into one .npz that ConvolutionalModel.restore_from_tf_arrays reads.

    python export_tf_checkpoint.py runs/2017-12-19T23h19m33s/model-epoch-004.chkpt exported.npz

Variable names contain '/', which numpy archive members may not: they are stored with '|' instead (the loader maps them back)."""
import sys

import numpy as np


def main(src, dst):
    import tensorflow as tf
    reader = tf.train.load_checkpoint(src)
    out = {}
    for name in reader.get_variable_to_shape_map():
        out[name.replace("/", "|")] = reader.get_tensor(name)
    np.savez(dst, **out)
    print("wrote %d variables to %s" % (len(out), dst))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
