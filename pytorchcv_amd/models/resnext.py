"""
    ResNeXt for ImageNet-1K on the MI355X hot path (reference pytorchcv/models/resnext.py:17-259): the grouped 3x3 (which
    carries the stride) runs as block-diagonal implicit GEMM over 32-channel group blocks.
"""

__all__ = ['ResNeXt', 'ResNeXtBottleneck', 'ResNeXtUnit', 'get_resnext']

import math
import torch.nn as nn
from .common.conv import conv1x1_block, conv3x3_block, conv_block_pair
from .resnet import ResInitBlock, ResStage
from ._build import ClassifierNet, add_stages, register_variants, stage_table
from ._tail import maybe_load_pretrained, DEFAULT_ROOT
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


class ResNeXt(ClassifierNet):
    def __init__(self, channels, init_block_channels, cardinality, bottleneck_width, in_channels=3, in_size=(224, 224),
                 num_classes=1000):
        super(ResNeXt, self).__init__(in_size, num_classes)
        self.features.add_module("init_block", ResInitBlock(in_channels=in_channels, out_channels=init_block_channels))
        width = add_stages(
            self.features, init_block_channels, channels, container=ResStage,
            make_unit=lambda cin, cout, stride, i, j: ResNeXtUnit(in_channels=cin, out_channels=cout, stride=stride,
                                                                  cardinality=cardinality, bottleneck_width=bottleneck_width))
        self.finish(width)


_DEPTHS = {14: (1, 1, 1, 1), 26: (2, 2, 2, 2), 38: (3, 3, 3, 3), 50: (3, 4, 6, 3), 101: (3, 4, 23, 3)}     # bottleneck units per stage


def get_resnext(blocks, cardinality, bottleneck_width, model_name=None, pretrained=False, root=DEFAULT_ROOT, **kwargs):
    if blocks not in _DEPTHS:
        raise ValueError("Unsupported ResNeXt with number of blocks: {}".format(blocks))
    net = ResNeXt(channels=stage_table((256, 512, 1024, 2048), _DEPTHS[blocks]), init_block_channels=64,
                  cardinality=cardinality, bottleneck_width=bottleneck_width, **kwargs)
    return maybe_load_pretrained(net, model_name, pretrained, root)


def _variant(name):
    """`resnext<blocks>_<cardinality>x<width>d` -> the three numbers."""
    depth, shape = name[len("resnext"):].split("_")
    card, width = shape[:-1].split("x")
    return dict(blocks=int(depth), cardinality=int(card), bottleneck_width=int(width))


register_variants(__name__, get_resnext, {n: _variant(n) for n in (
    "resnext14_16x4d", "resnext14_32x2d", "resnext14_32x4d", "resnext26_32x4d", "resnext50_32x4d", "resnext101_32x4d",
    "resnext101_64x4d")})
