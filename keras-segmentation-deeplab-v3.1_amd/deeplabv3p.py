"""Deeplabv3() — drop-in for the reference constructor (deeplabv3p.py:209-466).

Same signature, defaults, exception types, layer names and weight layout; the graph is declared
with the host-side layer mirror in graph.py and executed by the HIP engine (engine.py ->
libdl3.so).  The builder is table-driven (block specs below) rather than a call-by-call script.

Reference quirks that are reproduced on purpose:
  G1  MobileNetV2 ignores the caller's OS and always runs at output stride 8 (deeplabv3p.py:316).
  G2  MobileNetV2's ASPP has only the image-pooling and 1x1 branches (deeplabv3p.py:389-404).
  G3  ASPP/decoder "3x3 atrous convs" are depthwise-separable (deeplabv3p.py:392-399,:47-84).
Reference crashes that are repaired to their one obvious intent:
  `layers.add` (deeplabv3p.py:147,:149) is Add; `get_file` (deeplabv3p.py:458,:462) cannot download
  here, so weights='pascal_voc' loads the bonlime file from $DL3_WEIGHTS_DIR (or ~/.keras/models).
"""
import math
import os

from . import graph as G
from .graph import (Activation, Add, AveragePooling2D, BatchNormalization, Concatenate, Conv2D, DepthwiseConv2D,
                    Dropout, Input, KTensor, Model, Prescale, ReLU6, Reshape, ResizeBilinear, ZeroPadding2D)

WEIGHTS_FILE_X = "deeplabv3_xception_tf_dim_ordering_tf_kernels.h5"      # deeplabv3p.py:42,:458
WEIGHTS_FILE_MOBILE = "deeplabv3_mobilenetv2_tf_dim_ordering_tf_kernels.h5"  # deeplabv3p.py:43,:462

# (filters, stride, expansion, skip_connection, rate) for block_id 0..16 — deeplabv3p.py:327-367
_MNV2 = (
    (16, 1, 1, False, 1), (24, 2, 6, False, 1), (24, 1, 6, True, 1), (32, 2, 6, False, 1), (32, 1, 6, True, 1),
    (32, 1, 6, True, 1), (64, 1, 6, False, 1), (64, 1, 6, True, 2), (64, 1, 6, True, 2), (64, 1, 6, True, 2),
    (96, 1, 6, False, 2), (96, 1, 6, True, 2), (96, 1, 6, True, 2), (160, 1, 6, False, 2), (160, 1, 6, True, 4),
    (160, 1, 6, True, 4), (320, 1, 6, False, 4),
)


def _make_divisible(v, divisor, min_value=None):
    """Channel rounding of MobileNetV2 (deeplabv3p.py:157-164)."""
    floor = divisor if min_value is None else min_value
    rounded = max(floor, int(v + divisor / 2) // divisor * divisor)
    return rounded + divisor if rounded < 0.9 * v else rounded


def _explicit_same(x, kernel_size, rate):
    """The 'right same padding' of deeplabv3p.py:63-69,:105-110 for strided convs."""
    k_eff = kernel_size + (kernel_size - 1) * (rate - 1)
    beg = (k_eff - 1) // 2
    return ZeroPadding2D((beg, k_eff - 1 - beg))(x)


def SepConv_BN(x, filters, prefix, stride=1, kernel_size=3, rate=1, depth_activation=False, epsilon=1e-3):
    """dw3x3 -> BN -> [ReLU] -> pw1x1 -> BN -> [ReLU]; leading ReLU when depth_activation is False
    (deeplabv3p.py:47-84)."""
    padding = "same"
    if stride != 1:
        x = _explicit_same(x, kernel_size, rate)
        padding = "valid"
    if not depth_activation:
        x = Activation("relu")(x)
    x = DepthwiseConv2D((kernel_size, kernel_size), strides=(stride, stride), dilation_rate=(rate, rate),
                        padding=padding, use_bias=False, name=prefix + "_depthwise")(x)
    x = BatchNormalization(name=prefix + "_depthwise_BN", epsilon=epsilon)(x)
    if depth_activation:
        x = Activation("relu")(x)
    x = Conv2D(filters, (1, 1), padding="same", use_bias=False, name=prefix + "_pointwise")(x)
    x = BatchNormalization(name=prefix + "_pointwise_BN", epsilon=epsilon)(x)
    if depth_activation:
        x = Activation("relu")(x)
    return x


def _conv2d_same(x, filters, prefix, stride=1, kernel_size=3, rate=1):
    """deeplabv3p.py:87-116"""
    if stride == 1:
        return Conv2D(filters, (kernel_size, kernel_size), strides=(1, 1), padding="same", use_bias=False,
                      dilation_rate=(rate, rate), name=prefix)(x)
    x = _explicit_same(x, kernel_size, rate)
    return Conv2D(filters, (kernel_size, kernel_size), strides=(stride, stride), padding="valid", use_bias=False,
                  dilation_rate=(rate, rate), name=prefix)(x)


def _xception_block(inputs, depth_list, prefix, skip_connection_type, stride, rate=1, depth_activation=False,
                    return_skip=False):
    """deeplabv3p.py:119-155"""
    residual, skip = inputs, None
    for i, depth in enumerate(depth_list):
        residual = SepConv_BN(residual, depth, "%s_separable_conv%d" % (prefix, i + 1),
                              stride=stride if i == 2 else 1, rate=rate, depth_activation=depth_activation)
        if i == 1:
            skip = residual
    if skip_connection_type == "conv":
        shortcut = _conv2d_same(inputs, depth_list[-1], prefix + "_shortcut", kernel_size=1, stride=stride)
        shortcut = BatchNormalization(name=prefix + "_shortcut_BN")(shortcut)
        outputs = Add()([residual, shortcut])
    elif skip_connection_type == "sum":
        outputs = Add()([residual, inputs])
    else:
        outputs = residual
    return (outputs, skip) if return_skip else outputs


def _inverted_res_block(inputs, expansion, stride, alpha, filters, block_id, skip_connection, rate=1):
    """deeplabv3p.py:167-206"""
    in_channels = inputs._keras_shape[-1]
    pointwise_filters = _make_divisible(int(filters * alpha), 8)
    prefix = "expanded_conv_%d_" % block_id if block_id else "expanded_conv_"
    x = inputs
    if block_id:
        x = Conv2D(expansion * in_channels, kernel_size=1, padding="same", use_bias=False, name=prefix + "expand")(x)
        x = BatchNormalization(epsilon=1e-3, momentum=0.999, name=prefix + "expand_BN")(x)
        x = ReLU6(name=prefix + "expand_relu")(x)
    x = DepthwiseConv2D(kernel_size=3, strides=stride, use_bias=False, padding="same", dilation_rate=(rate, rate),
                        name=prefix + "depthwise")(x)
    x = BatchNormalization(epsilon=1e-3, momentum=0.999, name=prefix + "depthwise_BN")(x)
    x = ReLU6(name=prefix + "depthwise_relu")(x)
    x = Conv2D(pointwise_filters, kernel_size=1, padding="same", use_bias=False, name=prefix + "project")(x)
    x = BatchNormalization(epsilon=1e-3, momentum=0.999, name=prefix + "project_BN")(x)
    if skip_connection:
        return Add(name=prefix + "add")([inputs, x])
    return x


def _xception_backbone(x, OS):
    if OS == 8:
        entry_block3_stride, middle_block_rate, exit_block_rates, atrous_rates = 1, 2, (2, 4), (12, 24, 36)
    else:
        entry_block3_stride, middle_block_rate, exit_block_rates, atrous_rates = 2, 1, (1, 2), (6, 12, 18)
    x = Conv2D(32, (3, 3), strides=(2, 2), name="entry_flow_conv1_1", use_bias=False, padding="same")(x)
    x = Activation("relu")(BatchNormalization(name="entry_flow_conv1_1_BN")(x))
    x = _conv2d_same(x, 64, "entry_flow_conv1_2", kernel_size=3, stride=1)
    x = Activation("relu")(BatchNormalization(name="entry_flow_conv1_2_BN")(x))
    x = _xception_block(x, [128] * 3, "entry_flow_block1", "conv", stride=2)
    x, skip1 = _xception_block(x, [256] * 3, "entry_flow_block2", "conv", stride=2, return_skip=True)
    x = _xception_block(x, [728] * 3, "entry_flow_block3", "conv", stride=entry_block3_stride)
    for i in range(16):
        x = _xception_block(x, [728] * 3, "middle_flow_unit_%d" % (i + 1), "sum", stride=1, rate=middle_block_rate)
    x = _xception_block(x, [728, 1024, 1024], "exit_flow_block1", "conv", stride=1, rate=exit_block_rates[0])
    x = _xception_block(x, [1536, 1536, 2048], "exit_flow_block2", "none", stride=1, rate=exit_block_rates[1],
                        depth_activation=True)
    return x, skip1, atrous_rates


def _mobilenetv2_backbone(x, alpha):
    first = _make_divisible(32 * alpha, 8)
    x = Conv2D(first, kernel_size=3, strides=(2, 2), padding="same", use_bias=False, name="Conv")(x)
    x = BatchNormalization(epsilon=1e-3, momentum=0.999, name="Conv_BN")(x)
    x = ReLU6()(x)
    for block_id, (filters, stride, expansion, skip, rate) in enumerate(_MNV2):
        x = _inverted_res_block(x, filters=filters, alpha=alpha, stride=stride, expansion=expansion,
                                block_id=block_id, skip_connection=skip, rate=rate)
    return x


def weights_path(backbone):
    fname = WEIGHTS_FILE_X if backbone == "xception" else WEIGHTS_FILE_MOBILE
    for d in (os.environ.get("DL3_WEIGHTS_DIR"), os.path.join(os.path.expanduser("~"), ".keras", "models"), "."):
        if d and os.path.exists(os.path.join(d, fname)):
            return os.path.join(d, fname)
    raise FileNotFoundError(
        "weights='pascal_voc' needs %s (the reference downloads it from the bonlime/keras-deeplab-v3-plus 1.1 "
        "release, deeplabv3p.py:42-43); no network here — put it in $DL3_WEIGHTS_DIR or ~/.keras/models, "
        "or pass weights=None" % fname)


def Deeplabv3(weights="pascal_voc", input_tensor=None, infer=False, input_shape=(512, 512, 3), classes=21,
              backbone="mobilenetv2", OS=16, alpha=1.):
    """Instantiates the DeepLabV3+ architecture (reference: deeplabv3p.py:209-466).

    Arguments, defaults and raised exceptions are the reference's.  Returns a graph.Model whose
    predict()/train_on_batch() run on libdl3.so.  Output: [B, H*W, classes] softmax probabilities,
    or [B, H, W, classes] when infer=True (deeplabv3p.py:440-444); input: raw 0-255 float pixels."""
    if weights not in {"pascal_voc", None}:
        raise ValueError("The `weights` argument should be either `None` (random initialization) or "
                         "`pascal_voc` (pre-trained on PASCAL VOC)")
    if backbone not in {"xception", "mobilenetv2"}:
        raise ValueError("The `backbone` argument should be either `xception`  or `mobilenetv2` ")

    if input_tensor is None:
        img_input = Input(shape=input_shape)
    elif isinstance(input_tensor, KTensor):
        img_input = input_tensor
    else:
        img_input = Input(tensor=input_tensor, shape=input_shape)

    x = Prescale()(img_input)
    skip1 = None
    if backbone == "xception":
        x, skip1, atrous_rates = _xception_backbone(x, OS)
    else:
        OS = 8  # deeplabv3p.py:316
        x = _mobilenetv2_backbone(x, alpha)

    fh, fw = int(math.ceil(input_shape[0] / OS)), int(math.ceil(input_shape[1] / OS))
    # image-level feature branch (deeplabv3p.py:375-382)
    b4 = AveragePooling2D(pool_size=(fh, fw))(x)
    b4 = Conv2D(256, (1, 1), padding="same", use_bias=False, name="image_pooling")(b4)
    b4 = BatchNormalization(name="image_pooling_BN", epsilon=1e-5)(b4)
    b4 = Activation("relu")(b4)
    b4 = ResizeBilinear((fh, fw))(b4)
    # 1x1 branch (deeplabv3p.py:385-387)
    b0 = Conv2D(256, (1, 1), padding="same", use_bias=False, name="aspp0")(x)
    b0 = BatchNormalization(name="aspp0_BN", epsilon=1e-5)(b0)
    b0 = Activation("relu", name="aspp0_activation")(b0)
    branches = [b4, b0]
    if backbone == "xception":
        branches += [SepConv_BN(x, 256, "aspp%d" % (i + 1), rate=r, depth_activation=True, epsilon=1e-5)
                     for i, r in enumerate(atrous_rates)]
    x = Concatenate()(branches)
    x = Conv2D(256, (1, 1), padding="same", use_bias=False, name="concat_projection")(x)
    x = BatchNormalization(name="concat_projection_BN", epsilon=1e-5)(x)
    x = Activation("relu")(x)
    x = Dropout(0.1)(x)

    if backbone == "xception":  # decoder (deeplabv3p.py:414-429)
        x = ResizeBilinear((int(math.ceil(input_shape[0] / 4)), int(math.ceil(input_shape[1] / 4))))(x)
        dec_skip1 = Conv2D(48, (1, 1), padding="same", use_bias=False, name="feature_projection0")(skip1)
        dec_skip1 = BatchNormalization(name="feature_projection0_BN", epsilon=1e-5)(dec_skip1)
        dec_skip1 = Activation("relu")(dec_skip1)
        x = Concatenate()([x, dec_skip1])
        x = SepConv_BN(x, 256, "decoder_conv0", depth_activation=True, epsilon=1e-5)
        x = SepConv_BN(x, 256, "decoder_conv1", depth_activation=True, epsilon=1e-5)

    last_layer_name = "logits_semantic" if classes == 21 else "custom_logits_semantic"
    x = Conv2D(classes, (1, 1), padding="same", name=last_layer_name)(x)
    x = ResizeBilinear((input_shape[0], input_shape[1]))(x)
    if not infer:
        x = Reshape((input_shape[0] * input_shape[1], classes))(x)
    x = Activation("softmax")(x)

    model = Model(img_input, x, name="deeplabv3p")
    if weights == "pascal_voc":
        model.load_weights(weights_path(backbone), by_name=True)
    return model
