"""
    SE-ResNet for ImageNet-1K on the MI355X hot path (reference pytorchcv/models/seresnet.py:17-258): ResNet bodies with an
    SEBlock between body and skip add; the channel scale, the add and the ReLU are one pass (pcv_se_scale).
"""

__all__ = ['SEResNet', 'seresnet10', 'seresnet18', 'seresnet26', 'seresnetbc26b', 'seresnetbc38b', 'seresnet50', 'seresnet50b',
           'seresnet101', 'seresnet101b', 'seresnet152', 'SEResUnit', 'get_seresnet']

import torch.nn as nn
from .common.conv import conv1x1_block
from .common.att import SEBlock
from .resnet import ResStage, ResBlock, ResBottleneck, ResInitBlock, resnet_layers
from ._tail import AvgPool2dNHWC, LinearHead, run_net, maybe_load_pretrained, init_conv_params, DEFAULT_ROOT
from .. import engine


class SEResUnit(nn.Module):
    def __init__(self, in_channels, out_channels, stride, bottleneck, conv1_stride):
        super(SEResUnit, self).__init__()
        self.resize_identity = (in_channels != out_channels) or (stride != 1)
        if bottleneck:
            self.body = ResBottleneck(in_channels=in_channels, out_channels=out_channels, stride=stride,
                                      conv1_stride=conv1_stride)
        else:
            self.body = ResBlock(in_channels=in_channels, out_channels=out_channels, stride=stride)
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


class SEResNet(nn.Module):
    def __init__(self, channels, init_block_channels, bottleneck, conv1_stride, in_channels=3, in_size=(224, 224),
                 num_classes=1000):
        super(SEResNet, self).__init__()
        self.in_size = in_size
        self.num_classes = num_classes
        self.features = nn.Sequential()
        self.features.add_module("init_block", ResInitBlock(in_channels=in_channels, out_channels=init_block_channels))
        in_channels = init_block_channels
        for i, channels_per_stage in enumerate(channels):
            stage = ResStage()
            for j, out_channels in enumerate(channels_per_stage):
                stride = 2 if (j == 0) and (i != 0) else 1
                stage.add_module("unit{}".format(j + 1), SEResUnit(in_channels=in_channels, out_channels=out_channels,
                                                                   stride=stride, bottleneck=bottleneck,
                                                                   conv1_stride=conv1_stride))
                in_channels = out_channels
            self.features.add_module("stage{}".format(i + 1), stage)
        self.features.add_module("final_pool", AvgPool2dNHWC(kernel_size=7, stride=1, fp32_out=True))
        self.output = LinearHead(in_features=in_channels, out_features=num_classes)
        init_conv_params(self)

    def forward(self, x):
        return run_net(self, x, self.output)


def get_seresnet(blocks, bottleneck=None, conv1_stride=True, model_name=None, pretrained=False, root=DEFAULT_ROOT, **kwargs):
    if bottleneck is None:
        bottleneck = (blocks >= 50)
    layers = resnet_layers(blocks, bottleneck, what="SE-ResNet")
    widths = [64, 128, 256, 512]
    if bottleneck:
        widths = [w * 4 for w in widths]
    channels = [[w] * n for (w, n) in zip(widths, layers)]
    net = SEResNet(channels=channels, init_block_channels=64, bottleneck=bottleneck, conv1_stride=conv1_stride, **kwargs)
    return maybe_load_pretrained(net, model_name, pretrained, root)


def seresnet10(**kwargs):
    return get_seresnet(blocks=10, model_name="seresnet10", **kwargs)


def seresnet18(**kwargs):
    return get_seresnet(blocks=18, model_name="seresnet18", **kwargs)


def seresnet26(**kwargs):
    return get_seresnet(blocks=26, bottleneck=False, model_name="seresnet26", **kwargs)


def seresnetbc26b(**kwargs):
    return get_seresnet(blocks=26, bottleneck=True, conv1_stride=False, model_name="seresnetbc26b", **kwargs)


def seresnetbc38b(**kwargs):
    return get_seresnet(blocks=38, bottleneck=True, conv1_stride=False, model_name="seresnetbc38b", **kwargs)


def seresnet50(**kwargs):
    return get_seresnet(blocks=50, model_name="seresnet50", **kwargs)


def seresnet50b(**kwargs):
    return get_seresnet(blocks=50, conv1_stride=False, model_name="seresnet50b", **kwargs)


def seresnet101(**kwargs):
    return get_seresnet(blocks=101, model_name="seresnet101", **kwargs)


def seresnet101b(**kwargs):
    return get_seresnet(blocks=101, conv1_stride=False, model_name="seresnet101b", **kwargs)


def seresnet152(**kwargs):
    return get_seresnet(blocks=152, model_name="seresnet152", **kwargs)
