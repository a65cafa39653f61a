"""
    SE-ResNeXt for ImageNet-1K on the MI355X hot path (reference pytorchcv/models/seresnext.py:17-271): ResNeXt bottleneck
    (grouped 3x3 as block-diagonal implicit GEMM) + SEBlock, with the channel scale, skip add and ReLU in one pass.
"""

__all__ = ['SEResNeXt', 'seresnext50_32x4d', 'seresnext101_32x4d', 'seresnext101_64x4d', 'SEResNeXtUnit', 'get_seresnext']

import torch.nn as nn
from .common.conv import conv1x1_block
from .common.att import SEBlock
from .resnet import ResStage, ResInitBlock
from .resnext import ResNeXtBottleneck
from ._tail import AvgPool2dNHWC, LinearHead, run_net, maybe_load_pretrained, init_conv_params, DEFAULT_ROOT
from .. import engine


class SEResNeXtUnit(nn.Module):
    def __init__(self, in_channels, out_channels, stride, cardinality, bottleneck_width):
        super(SEResNeXtUnit, self).__init__()
        self.resize_identity = (in_channels != out_channels) or (stride != 1)
        self.body = ResNeXtBottleneck(in_channels=in_channels, out_channels=out_channels, stride=stride,
                                      cardinality=cardinality, bottleneck_width=bottleneck_width)
        self.se = SEBlock(channels=out_channels)
        if self.resize_identity:
            self.identity_conv = conv1x1_block(in_channels=in_channels, out_channels=out_channels, stride=stride,
                                               activation=None)
        self.activ = nn.ReLU(inplace=True)

    pcv_chainable = True         # ResStage: first convolution handed in, last one (with the SE block) fused forward

    def run_chained(self, a, conv1_out=None, next_unit=None):
        identity = self.identity_conv(a) if self.resize_identity else a
        body = self.body
        if not hasattr(body, "conv3"):                           # basic block: the SE block follows a 3x3 (not affine in the mean)
            return self.se(body(a), residual=identity, post_act=self.activ), None
        z = body.conv2(conv1_out if conv1_out is not None else body.conv1(a))
        nxt = next_unit.body.conv1 if (next_unit is not None and hasattr(next_unit.body, "conv3")) else None
        y = self.se.run_behind(body.conv3, z, residual=identity, post_act=self.activ, next_conv=nxt)
        if y is None:
            y = self.se(body.conv3(z), residual=identity, post_act=self.activ)
        return y if isinstance(y, tuple) else (y, None)

    def _run(self, a):
        return self.run_chained(a)[0]

    def forward(self, x):
        return engine.boundary(self, x, self._run)


class SEResNeXt(nn.Module):
    def __init__(self, channels, init_block_channels, cardinality, bottleneck_width, in_channels=3, in_size=(224, 224),
                 num_classes=1000):
        super(SEResNeXt, self).__init__()
        self.in_size = in_size
        self.num_classes = num_classes
        self.features = nn.Sequential()
        self.features.add_module("init_block", ResInitBlock(in_channels=in_channels, out_channels=init_block_channels))
        in_channels = init_block_channels
        for i, channels_per_stage in enumerate(channels):
            stage = ResStage()
            for j, out_channels in enumerate(channels_per_stage):
                stride = 2 if (j == 0) and (i != 0) else 1
                stage.add_module("unit{}".format(j + 1), SEResNeXtUnit(in_channels=in_channels, out_channels=out_channels,
                                                                       stride=stride, cardinality=cardinality,
                                                                       bottleneck_width=bottleneck_width))
                in_channels = out_channels
            self.features.add_module("stage{}".format(i + 1), stage)
        self.features.add_module("final_pool", AvgPool2dNHWC(kernel_size=7, stride=1, fp32_out=True))
        self.output = LinearHead(in_features=in_channels, out_features=num_classes)
        init_conv_params(self)

    def forward(self, x):
        return run_net(self, x, self.output)


def get_seresnext(blocks, cardinality, bottleneck_width, model_name=None, pretrained=False, root=DEFAULT_ROOT, **kwargs):
    table = {50: [3, 4, 6, 3], 101: [3, 4, 23, 3]}
    if blocks not in table:
        raise ValueError("Unsupported SE-ResNeXt with number of blocks: {}".format(blocks))
    channels = [[w] * n for (w, n) in zip([256, 512, 1024, 2048], table[blocks])]
    net = SEResNeXt(channels=channels, init_block_channels=64, cardinality=cardinality, bottleneck_width=bottleneck_width,
                    **kwargs)
    return maybe_load_pretrained(net, model_name, pretrained, root)


def seresnext50_32x4d(**kwargs):
    return get_seresnext(blocks=50, cardinality=32, bottleneck_width=4, model_name="seresnext50_32x4d", **kwargs)


def seresnext101_32x4d(**kwargs):
    return get_seresnext(blocks=101, cardinality=32, bottleneck_width=4, model_name="seresnext101_32x4d", **kwargs)


def seresnext101_64x4d(**kwargs):
    return get_seresnext(blocks=101, cardinality=64, bottleneck_width=4, model_name="seresnext101_64x4d", **kwargs)
