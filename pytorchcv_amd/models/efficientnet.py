"""
    EfficientNet for ImageNet-1K on the MI355X hot path (reference pytorchcv/models/efficientnet.py:27-1230): MBConv units =
    expand 1x1 (MFMA) -> depthwise 3x3/5x5 -> SE (swish inside the excitation) -> project 1x1 with the skip add in its
    epilogue; swish is an epilogue code. TF-"same" mode (the `b`/`c` weights) pads asymmetrically per input size: the pad
    the reference applies with F.pad becomes the four pad fields of the launch, no padded copy is made.
"""

__all__ = ['EfficientNet', 'calc_tf_padding', 'EffiInvResUnit', 'EffiInitBlock', 'EffiDwsConvUnit', 'get_efficientnet']

import math
import torch.nn as nn
from .common.activ import lambda_swish
from .common.norm import lambda_batchnorm2d
from .common.conv import conv1x1_block, conv3x3_block, dwconv3x3_block, dwconv5x5_block
from .common.att import round_channels, SEBlock
from ._tail import GlobalAvgPool2dNHWC, LinearHead, run_net, maybe_load_pretrained, init_conv_params, DEFAULT_ROOT
from .. import engine


def calc_tf_padding(x, kernel_size, stride=1, dilation=1):
    """TF-"same" padding of reference efficientnet.py:27-55 for an NCHW tensor or an NHWC handle. The tuple goes to
    F.pad in the reference, i.e. it is read as (left, right, top, bottom) although it is computed (h, h, w, w); the two
    only differ on non-square maps and the quirk is kept so that results stay identical."""
    if isinstance(x, engine.NHWC):
        height, width = x.H, x.W
    else:
        height, width = x.size()[2:]
    oh = math.ceil(float(height) / stride)
    ow = math.ceil(float(width) / stride)
    pad_h = max((oh - 1) * stride + (kernel_size - 1) * dilation + 1 - height, 0)
    pad_w = max((ow - 1) * stride + (kernel_size - 1) * dilation + 1 - width, 0)
    return pad_h // 2, pad_h - pad_h // 2, pad_w // 2, pad_w - pad_w // 2


class EffiDwsConvUnit(nn.Module):
    """Depthwise-separable first-stage unit (reference efficientnet.py:58-115). `stride` is accepted and, as in the
    reference, not used by the depthwise convolution."""
    def __init__(self, in_channels, out_channels, stride, normalization, activation, tf_mode):
        super(EffiDwsConvUnit, self).__init__()
        self.tf_mode = tf_mode
        self.residual = (in_channels == out_channels) and (stride == 1)
        self.dw_conv = dwconv3x3_block(in_channels=in_channels, out_channels=in_channels, padding=(0 if tf_mode else 1),
                                       normalization=normalization, activation=activation)
        self.se = SEBlock(channels=in_channels, reduction=4, mid_activation=activation)
        self.pw_conv = conv1x1_block(in_channels=in_channels, out_channels=out_channels, normalization=normalization,
                                     activation=None)

    def _run(self, a):
        pad4 = calc_tf_padding(a, kernel_size=3) if self.tf_mode else None
        y = self.se(self.dw_conv(a, pad4=pad4))
        return self.pw_conv(y, residual=(a if self.residual else None))

    def forward(self, x):
        return engine.boundary(self, x, self._run)


class EffiInvResUnit(nn.Module):
    """Inverted residual (MBConv) unit (reference efficientnet.py:118-197)."""
    def __init__(self, in_channels, out_channels, kernel_size, stride, exp_factor, se_factor, normalization, activation,
                 tf_mode):
        super(EffiInvResUnit, self).__init__()
        self.kernel_size = kernel_size
        self.stride = stride
        self.tf_mode = tf_mode
        self.residual = (in_channels == out_channels) and (stride == 1)
        self.use_se = se_factor > 0
        mid_channels = in_channels * exp_factor
        dwconv_block_fn = dwconv3x3_block if kernel_size == 3 else (dwconv5x5_block if kernel_size == 5 else None)
        self.conv1 = conv1x1_block(in_channels=in_channels, out_channels=mid_channels, normalization=normalization,
                                   activation=activation)
        self.conv2 = dwconv_block_fn(in_channels=mid_channels, out_channels=mid_channels, stride=stride,
                                     padding=(0 if tf_mode else (kernel_size // 2)), normalization=normalization,
                                     activation=activation)
        if self.use_se:
            self.se = SEBlock(channels=mid_channels, reduction=(exp_factor * se_factor), mid_activation=activation)
        self.conv3 = conv1x1_block(in_channels=mid_channels, out_channels=out_channels, normalization=normalization,
                                   activation=None)

    def _run(self, a):
        y = self.conv1(a)
        pad4 = calc_tf_padding(y, kernel_size=self.kernel_size, stride=self.stride) if self.tf_mode else None
        y = self.conv2(y, pad4=pad4)
        if self.use_se:
            y = self.se(y)
        return self.conv3(y, residual=(a if self.residual else None))

    def forward(self, x):
        return engine.boundary(self, x, self._run)


class EffiInitBlock(nn.Module):
    """3x3/2 stem (reference efficientnet.py:200-239)."""
    def __init__(self, in_channels, out_channels, normalization, activation, tf_mode):
        super(EffiInitBlock, self).__init__()
        self.tf_mode = tf_mode
        self.conv = conv3x3_block(in_channels=in_channels, out_channels=out_channels, stride=2,
                                  padding=(0 if tf_mode else 1), normalization=normalization, activation=activation)

    def _run(self, a):
        return self.conv(a, pad4=(calc_tf_padding(a, kernel_size=3, stride=2) if self.tf_mode else None))

    def forward(self, x):
        return engine.boundary(self, x, self._run, stem=True)


class EfficientNet(nn.Module):
    pcv_16bit = "fp16"      # the 16-bit mode "auto" resolves to for this family (engine.compute_dtype_of)

    def __init__(self, channels, init_block_channels, final_block_channels, kernel_sizes, strides_per_stage,
                 expansion_factors, dropout_rate=0.2, tf_mode=False, bn_eps=1e-5, in_channels=3, in_size=(224, 224),
                 num_classes=1000):
        super(EfficientNet, self).__init__()
        self.in_size = in_size
        self.num_classes = num_classes
        normalization = lambda_batchnorm2d(eps=bn_eps)
        activation = lambda_swish()
        self.features = nn.Sequential()
        self.features.add_module("init_block", EffiInitBlock(in_channels=in_channels, out_channels=init_block_channels,
                                                             normalization=normalization, activation=activation,
                                                             tf_mode=tf_mode))
        in_channels = init_block_channels
        for i, channels_per_stage in enumerate(channels):
            stage = nn.Sequential()
            for j, out_channels in enumerate(channels_per_stage):
                stride = strides_per_stage[i] if (j == 0) else 1
                if i == 0:
                    unit = EffiDwsConvUnit(in_channels=in_channels, out_channels=out_channels, stride=stride,
                                           normalization=normalization, activation=activation, tf_mode=tf_mode)
                else:
                    unit = EffiInvResUnit(in_channels=in_channels, out_channels=out_channels,
                                          kernel_size=kernel_sizes[i][j], stride=stride, exp_factor=expansion_factors[i][j],
                                          se_factor=4, normalization=normalization, activation=activation, tf_mode=tf_mode)
                stage.add_module("unit{}".format(j + 1), unit)
                in_channels = out_channels
            self.features.add_module("stage{}".format(i + 1), stage)
        self.features.add_module("final_block", conv1x1_block(in_channels=in_channels, out_channels=final_block_channels,
                                                              normalization=normalization, activation=activation))
        in_channels = final_block_channels
        self.features.add_module("final_pool", GlobalAvgPool2dNHWC(output_size=1, fp32_out=True))
        self.output = nn.Sequential()
        if dropout_rate > 0.0:
            self.output.add_module("dropout", nn.Dropout(p=dropout_rate))      # identity at inference: never launched
        self.output.add_module("fc", LinearHead(in_features=in_channels, out_features=num_classes))
        init_conv_params(self)
        engine.stamp_family_dtype(self)                    # sub-modules called on their own resolve "auto" like the net

    def _head(self, a):
        if self.training:
            raise RuntimeError("EfficientNet: the MI355X path is inference only (call net.eval())")
        return self.output.fc(a)

    def forward(self, x):
        return run_net(self, x, self._head)


# version -> (in_size, depth_factor, width_factor, dropout_rate), reference efficientnet.py:393-440
_VERSIONS = {"b0": (224, 1.0, 1.0, 0.2), "b1": (240, 1.1, 1.0, 0.2), "b2": (260, 1.2, 1.1, 0.3), "b3": (300, 1.4, 1.2, 0.3),
             "b4": (380, 1.8, 1.4, 0.4), "b5": (456, 2.2, 1.6, 0.4), "b6": (528, 2.6, 1.8, 0.5), "b7": (600, 3.1, 2.0, 0.5),
             "b8": (672, 3.6, 2.2, 0.5)}


def _by_stage(values, layers, downsample):
    """Expand per-layer values into per-stage lists: a new stage wherever `downsample` is set."""
    out = []
    for value, count, new_stage in zip(values, layers, downsample):
        if new_stage != 0:
            out.append([value] * count)
        else:
            out[-1] = out[-1] + [value] * count
    return out


def get_efficientnet(version, in_size, tf_mode=False, bn_eps=1e-5, model_name=None, pretrained=False, root=DEFAULT_ROOT,
                     **kwargs):
    if version not in _VERSIONS:
        raise ValueError("Unsupported EfficientNet version {}".format(version))
    size, depth_factor, width_factor, dropout_rate = _VERSIONS[version]
    assert (tuple(in_size) == (size, size))
    downsample = [1, 1, 1, 1, 0, 1, 0]
    layers = [int(math.ceil(li * depth_factor)) for li in [1, 2, 2, 3, 3, 4, 1]]
    channels_per_layers = [round_channels(ci * width_factor) for ci in [16, 24, 40, 80, 112, 192, 320]]
    channels = _by_stage(channels_per_layers, layers, downsample)
    kernel_sizes = _by_stage([3, 3, 5, 3, 5, 5, 3], layers, downsample)
    expansion_factors = _by_stage([1, 6, 6, 6, 6, 6, 6], layers, downsample)
    strides_per_stage = [si[0] for si in _by_stage([1, 2, 2, 2, 1, 2, 1], layers, downsample)]
    init_block_channels = round_channels(32 * width_factor)
    final_block_channels = 1280
    if width_factor > 1.0:
        assert (int(final_block_channels * width_factor) == round_channels(final_block_channels * width_factor))
        final_block_channels = round_channels(final_block_channels * width_factor)
    net = EfficientNet(channels=channels, init_block_channels=init_block_channels, final_block_channels=final_block_channels,
                       kernel_sizes=kernel_sizes, strides_per_stage=strides_per_stage, expansion_factors=expansion_factors,
                       dropout_rate=dropout_rate, tf_mode=tf_mode, bn_eps=bn_eps, in_size=in_size, **kwargs)
    return maybe_load_pretrained(net, model_name, pretrained, root)


def _variant(version, suffix):
    name = "efficientnet_{}{}".format(version, suffix)
    size = _VERSIONS[version][0]
    tf = dict(tf_mode=True, bn_eps=1e-3) if suffix else {}

    def factory(in_size=(size, size), **kwargs):
        return get_efficientnet(version=version, in_size=in_size, model_name=name, **dict(tf, **kwargs))
    factory.__name__ = name
    factory.__doc__ = "EfficientNet-{}{} (reference efficientnet.py:496-1230).".format(
        version.upper(), " with TF-same padding and BN eps 1e-3 ('{}' weights)".format(suffix) if suffix else "")
    return factory


for _v in _VERSIONS:
    for _s in ("", "b", "c"):
        if _v == "b8" and _s == "b":
            continue                                            # the reference has no efficientnet_b8b
        globals()["efficientnet_{}{}".format(_v, _s)] = _variant(_v, _s)
        __all__.append("efficientnet_{}{}".format(_v, _s))
