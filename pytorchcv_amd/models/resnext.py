"""
    ResNeXt for ImageNet-1K on the MI355X hot path (reference pytorchcv/models/resnext.py:17-259): the grouped 3x3 (which
    carries the stride) runs as block-diagonal implicit GEMM over 32-channel group blocks.
"""

__all__ = ['ResNeXt', 'resnext14_16x4d', 'resnext14_32x2d', 'resnext14_32x4d', 'resnext26_32x4d', 'resnext50_32x4d',
           'resnext101_32x4d', 'resnext101_64x4d', 'ResNeXtBottleneck', 'ResNeXtUnit', 'get_resnext']

import math
import torch.nn as nn
from .common.conv import conv1x1_block, conv3x3_block, conv_block_pair
from .resnet import ResInitBlock, ResStage
from ._tail import AvgPool2dNHWC, LinearHead, run_net, maybe_load_pretrained, init_conv_params, DEFAULT_ROOT
from .. import engine


class ResNeXtBottleneck(nn.Module):
    def __init__(self, in_channels, out_channels, stride, cardinality, bottleneck_width, bottleneck_factor=4):
        super(ResNeXtBottleneck, self).__init__()
        mid_channels = out_channels // bottleneck_factor
        D = int(math.floor(mid_channels * (bottleneck_width / 64.0)))
        group_width = cardinality * D
        self.conv1 = conv1x1_block(in_channels=in_channels, out_channels=group_width)
        self.conv2 = conv3x3_block(in_channels=group_width, out_channels=group_width, stride=stride, groups=cardinality)
        self.conv3 = conv1x1_block(in_channels=group_width, out_channels=out_channels, activation=None)

    def forward(self, x, residual=None, post_act=None):
        return engine.boundary(self, x, lambda a: self.conv3(self.conv2(self.conv1(a)), residual=residual, post_act=post_act))


class ResNeXtUnit(nn.Module):
    def __init__(self, in_channels, out_channels, stride, cardinality, bottleneck_width):
        super(ResNeXtUnit, self).__init__()
        self.resize_identity = (in_channels != out_channels) or (stride != 1)
        self.body = ResNeXtBottleneck(in_channels=in_channels, out_channels=out_channels, stride=stride,
                                      cardinality=cardinality, bottleneck_width=bottleneck_width)
        if self.resize_identity:
            self.identity_conv = conv1x1_block(in_channels=in_channels, out_channels=out_channels, stride=stride,
                                               activation=None)
        self.activ = nn.ReLU(inplace=True)

    pcv_chainable = True         # ResStage may hand this unit its first convolution's output and fuse its last one forward

    def _run(self, a):
        identity = self.identity_conv(a) if self.resize_identity else a
        return self.body(a, residual=identity, post_act=self.activ)

    def run_chained(self, a, conv1_out=None, next_unit=None):
        """As ResUnit.run_chained: the unit's last 1x1 (+ skip add + ReLU) and the next unit's first 1x1 as one launch when the
        pair is covered (128 -> 256 -> 128 and 256 -> 512 -> 256 in the 32x4d nets)."""
        identity = self.identity_conv(a) if self.resize_identity else a
        body = self.body
        y = body.conv2(conv1_out if conv1_out is not None else body.conv1(a))
        if next_unit is not None:
            pair = conv_block_pair(body.conv3, y, identity, self.activ, next_unit.body.conv1)
            if pair is not None:
                return pair
        return body.conv3(y, residual=identity, post_act=self.activ), None

    def forward(self, x):
        return engine.boundary(self, x, self._run)


class ResNeXt(nn.Module):
    def __init__(self, channels, init_block_channels, cardinality, bottleneck_width, in_channels=3, in_size=(224, 224),
                 num_classes=1000):
        super(ResNeXt, self).__init__()
        self.in_size = in_size
        self.num_classes = num_classes
        self.features = nn.Sequential()
        self.features.add_module("init_block", ResInitBlock(in_channels=in_channels, out_channels=init_block_channels))
        in_channels = init_block_channels
        for i, channels_per_stage in enumerate(channels):
            stage = ResStage()
            for j, out_channels in enumerate(channels_per_stage):
                stride = 2 if (j == 0) and (i != 0) else 1
                stage.add_module("unit{}".format(j + 1), ResNeXtUnit(in_channels=in_channels, out_channels=out_channels,
                                                                     stride=stride, cardinality=cardinality,
                                                                     bottleneck_width=bottleneck_width))
                in_channels = out_channels
            self.features.add_module("stage{}".format(i + 1), stage)
        self.features.add_module("final_pool", AvgPool2dNHWC(kernel_size=7, stride=1))
        self.output = LinearHead(in_features=in_channels, out_features=num_classes)
        init_conv_params(self)

    def forward(self, x):
        return run_net(self, x, self.output)


def get_resnext(blocks, cardinality, bottleneck_width, model_name=None, pretrained=False, root=DEFAULT_ROOT, **kwargs):
    table = {14: [1, 1, 1, 1], 26: [2, 2, 2, 2], 38: [3, 3, 3, 3], 50: [3, 4, 6, 3], 101: [3, 4, 23, 3]}
    if blocks not in table:
        raise ValueError("Unsupported ResNeXt with number of blocks: {}".format(blocks))
    layers = table[blocks]
    assert (sum(layers) * 3 + 2 == blocks)
    channels = [[w] * n for (w, n) in zip([256, 512, 1024, 2048], layers)]
    net = ResNeXt(channels=channels, init_block_channels=64, cardinality=cardinality, bottleneck_width=bottleneck_width,
                  **kwargs)
    return maybe_load_pretrained(net, model_name, pretrained, root)


def resnext14_16x4d(**kwargs):
    return get_resnext(blocks=14, cardinality=16, bottleneck_width=4, model_name="resnext14_16x4d", **kwargs)


def resnext14_32x2d(**kwargs):
    return get_resnext(blocks=14, cardinality=32, bottleneck_width=2, model_name="resnext14_32x2d", **kwargs)


def resnext14_32x4d(**kwargs):
    return get_resnext(blocks=14, cardinality=32, bottleneck_width=4, model_name="resnext14_32x4d", **kwargs)


def resnext26_32x4d(**kwargs):
    return get_resnext(blocks=26, cardinality=32, bottleneck_width=4, model_name="resnext26_32x4d", **kwargs)


def resnext50_32x4d(**kwargs):
    return get_resnext(blocks=50, cardinality=32, bottleneck_width=4, model_name="resnext50_32x4d", **kwargs)


def resnext101_32x4d(**kwargs):
    return get_resnext(blocks=101, cardinality=32, bottleneck_width=4, model_name="resnext101_32x4d", **kwargs)


def resnext101_64x4d(**kwargs):
    return get_resnext(blocks=101, cardinality=64, bottleneck_width=4, model_name="resnext101_64x4d", **kwargs)
