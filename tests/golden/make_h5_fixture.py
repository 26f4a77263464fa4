"""Writes tests/golden/keras_style_h5py.h5 with REAL h5py/libhdf5 (run with an interpreter that has h5py, e.g.
/opt/conda/bin/python3.9 here) in the layout Keras 2.2.x `save_weights` produces: the fixture pins
the package's own HDF5 reader (h5lite.py) against libhdf5's byte layout.  Values are deterministic: arange-based."""
import os

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
layers = [("input_1", []), ("lambda_1", []),
          ("Conv", [("Conv/kernel:0", (3, 3, 3, 8))]),
          ("Conv_BN", [("Conv_BN/gamma:0", (8,)), ("Conv_BN/beta:0", (8,)), ("Conv_BN/moving_mean:0", (8,)),
                       ("Conv_BN/moving_variance:0", (8,))]),
          ("expanded_conv_depthwise", [("expanded_conv_depthwise/depthwise_kernel:0", (3, 3, 8, 1))]),
          ("logits_semantic", [("logits_semantic/kernel:0", (1, 1, 8, 21)), ("logits_semantic/bias:0", (21,))])]
layers += [("middle_flow_unit_%d_separable_conv1_pointwise" % i,
            [("middle_flow_unit_%d_separable_conv1_pointwise/kernel:0" % i, (1, 1, 2, 3))]) for i in range(1, 41)]


def value(name, shape):
    n = int(np.prod(shape))
    return ((np.arange(n, dtype=np.float32) * 0.25 + len(name)) % 7.0 - 3.0).reshape(shape)


with h5py.File(os.path.join(HERE, "keras_style_h5py.h5"), "w", libver="earliest") as f:
    f.attrs["layer_names"] = np.array([n.encode() for n, _ in layers], dtype="S")  # h5py 2.x / Keras 2.2.4 style
    f.attrs["backend"] = b"tensorflow"
    f.attrs["keras_version"] = b"2.2.4"
    for n, ws in layers:
        g = f.create_group(n)
        g.attrs["weight_names"] = np.array([w.encode() for w, _ in ws], dtype="S") if ws else np.zeros((0,), "S1")
        for w, shp in ws:
            d = g.create_dataset(w, shp, dtype="float32")
            d[...] = value(w, shp)
print("written", len(layers), "layers")
