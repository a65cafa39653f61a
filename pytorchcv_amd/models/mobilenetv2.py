"""
    MobileNetV2 for ImageNet-1K on the MI355X hot path (reference pytorchcv/models/mobilenetv2.py:16-220): expand 1x1 (MFMA),
    depthwise 3x3 (direct HBM-bound kernel), project 1x1 with the skip add in its epilogue; bias-free 1x1-conv classifier.
"""

__all__ = ['MobileNetV2', 'mobilenetv2_w1', 'mobilenetv2_w3d4', 'mobilenetv2_wd2', 'mobilenetv2_wd4', 'LinearBottleneck',
           'get_mobilenetv2']

import torch.nn as nn
from .common.activ import lambda_relu6
from .common.conv import conv1x1, conv1x1_block, conv3x3_block, dwconv3x3_block, mbconv_chain
from ._tail import AvgPool2dNHWC, run_net, maybe_load_pretrained, init_conv_params, DEFAULT_ROOT
from .. import engine


class LinearBottleneck(nn.Module):
    """Inverted residual unit (reference mobilenetv2.py:16-71); no activation after the add."""
    def __init__(self, in_channels, out_channels, stride, expansion, remove_exp_conv, activation):
        super(LinearBottleneck, self).__init__()
        self.residual = (in_channels == out_channels) and (stride == 1)
        mid_channels = in_channels * 6 if expansion else in_channels
        self.use_exp_conv = (expansion or (not remove_exp_conv))
        if self.use_exp_conv:
            self.conv1 = conv1x1_block(in_channels=in_channels, out_channels=mid_channels, activation=activation)
        self.conv2 = dwconv3x3_block(in_channels=mid_channels, out_channels=mid_channels, stride=stride, activation=activation)
        self.conv3 = conv1x1_block(in_channels=mid_channels, out_channels=out_channels, activation=None)

    def _run(self, a):
        residual = a if self.residual else None
        y = mbconv_chain(self.conv1 if self.use_exp_conv else None, self.conv2, self.conv3, a, residual=residual)
        if y is not None:
            return y                                     # the whole unit was one launch (csrc/mbconv.hpp)
        y = self.conv1(a) if self.use_exp_conv else a
        return self.conv3(self.conv2(y), residual=residual)

    def forward(self, x):
        return engine.boundary(self, x, self._run)


class MobileNetV2(nn.Module):
    def __init__(self, channels, init_block_channels, final_block_channels, remove_exp_conv, in_channels=3,
                 in_size=(224, 224), num_classes=1000):
        super(MobileNetV2, self).__init__()
        self.in_size = in_size
        self.num_classes = num_classes
        activation = lambda_relu6()
        self.features = nn.Sequential()
        self.features.add_module("init_block", conv3x3_block(in_channels=in_channels, out_channels=init_block_channels,
                                                             stride=2, activation=activation))
        in_channels = init_block_channels
        for i, channels_per_stage in enumerate(channels):
            stage = nn.Sequential()
            for j, out_channels in enumerate(channels_per_stage):
                stride = 2 if (j == 0) and (i != 0) else 1
                expansion = (i != 0) or (j != 0)
                stage.add_module("unit{}".format(j + 1), LinearBottleneck(
                    in_channels=in_channels, out_channels=out_channels, stride=stride, expansion=expansion,
                    remove_exp_conv=remove_exp_conv, activation=activation))
                in_channels = out_channels
            self.features.add_module("stage{}".format(i + 1), stage)
        self.features.add_module("final_block", conv1x1_block(in_channels=in_channels, out_channels=final_block_channels,
                                                              activation=activation))
        in_channels = final_block_channels
        self.features.add_module("final_pool", AvgPool2dNHWC(kernel_size=7, stride=1))
        self.output = conv1x1(in_channels=in_channels, out_channels=num_classes, bias=False)
        init_conv_params(self)

    def _head(self, a):
        if a.H != 1 or a.W != 1:
            raise RuntimeError("classifier expects a 1x1 pooled map, got {}x{}".format(a.H, a.W))
        y = self.output(a, out_fp32=True)
        return y.t.view(y.N, -1)

    def forward(self, x):
        return run_net(self, x, self._head)


def get_mobilenetv2(width_scale, remove_exp_conv=False, model_name=None, pretrained=False, root=DEFAULT_ROOT, **kwargs):
    """Channel plan of reference mobilenetv2.py:183-203: a stage boundary wherever `downsample` is set."""
    init_block_channels, final_block_channels = 32, 1280
    plan = [(16, 1, 0), (24, 2, 1), (32, 3, 1), (64, 4, 1), (96, 3, 0), (160, 3, 1), (320, 1, 0)]
    channels = [[]]
    for width, count, downsample in plan:
        if downsample:
            channels.append([width] * count)
        else:
            channels[-1] = channels[-1] + [width] * count
    if width_scale != 1.0:
        channels = [[int(c * width_scale) for c in ci] for ci in channels]
        init_block_channels = int(init_block_channels * width_scale)
        if width_scale > 1.0:
            final_block_channels = int(final_block_channels * width_scale)
    net = MobileNetV2(channels=channels, init_block_channels=init_block_channels, final_block_channels=final_block_channels,
                      remove_exp_conv=remove_exp_conv, **kwargs)
    return maybe_load_pretrained(net, model_name, pretrained, root)


def mobilenetv2_w1(**kwargs):
    return get_mobilenetv2(width_scale=1.0, model_name="mobilenetv2_w1", **kwargs)


def mobilenetv2_w3d4(**kwargs):
    return get_mobilenetv2(width_scale=0.75, model_name="mobilenetv2_w3d4", **kwargs)


def mobilenetv2_wd2(**kwargs):
    return get_mobilenetv2(width_scale=0.5, model_name="mobilenetv2_wd2", **kwargs)


def mobilenetv2_wd4(**kwargs):
    return get_mobilenetv2(width_scale=0.25, model_name="mobilenetv2_wd4", **kwargs)
