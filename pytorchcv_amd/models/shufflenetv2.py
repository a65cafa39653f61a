"""
    ShuffleNetV2 for ImageNet-1K on the MI355X hot path (reference pytorchcv/models/shufflenetv2.py:17-290). The unit's channel
    split, concatenation and channel shuffle are two data-movement launches (pcv_channel_slice, pcv_channel_interleave2: cat +
    shuffle(groups=2) is an interleave); its 1x1 / depthwise / 1x1 branch are fused launches with the stand-alone BatchNorm
    modules of the reference (`compress_bn1`, `dw_bn2`, ...) folded into their epilogues. Widths such as 58 or 116 channels run
    on zero-padded physical channel counts (engine.ConvRunner).
"""

__all__ = ['ShuffleNetV2', 'shufflenetv2_wd2', 'shufflenetv2_w1', 'shufflenetv2_w3d2', 'shufflenetv2_w2', 'ShuffleUnit',
           'ShuffleInitBlock', 'get_shufflenetv2']

import torch.nn as nn
from .common.conv import conv1x1, depthwise_conv3x3, conv1x1_block, conv3x3_block
from .common.att import SEBlock
from ._tail import MaxPool2dNHWC, AvgPool2dNHWC, LinearHead, run_net, maybe_load_pretrained, init_conv_params, DEFAULT_ROOT
from .. import engine


class ChannelShuffle(nn.Module):
    """Parameter-free marker with the reference's constructor (common/tutti.py:294-321); the shuffle itself is fused with the
    concatenation that precedes it (engine.cat_shuffle2)."""
    def __init__(self, channels, groups):
        super(ChannelShuffle, self).__init__()
        if channels % groups != 0:
            raise ValueError("channels must be divisible by groups")
        self.groups = groups


class ShuffleUnit(nn.Module):
    """reference shufflenetv2.py:17-91."""
    def __init__(self, in_channels, out_channels, downsample, use_se, use_residual):
        super(ShuffleUnit, self).__init__()
        self.downsample = downsample
        self.use_se = use_se
        self.use_residual = use_residual
        mid_channels = out_channels // 2
        self.mid_channels = mid_channels
        self.compress_conv1 = conv1x1(in_channels=(in_channels if self.downsample else mid_channels), out_channels=mid_channels)
        self.compress_bn1 = nn.BatchNorm2d(num_features=mid_channels)
        self.dw_conv2 = depthwise_conv3x3(channels=mid_channels, stride=(2 if self.downsample else 1))
        self.dw_bn2 = nn.BatchNorm2d(num_features=mid_channels)
        self.expand_conv3 = conv1x1(in_channels=mid_channels, out_channels=mid_channels)
        self.expand_bn3 = nn.BatchNorm2d(num_features=mid_channels)
        if self.use_se:
            self.se = SEBlock(channels=mid_channels)
        if downsample:
            self.dw_conv4 = depthwise_conv3x3(channels=in_channels, stride=2)
            self.dw_bn4 = nn.BatchNorm2d(num_features=in_channels)
            self.expand_conv5 = conv1x1(in_channels=in_channels, out_channels=mid_channels)
            self.expand_bn5 = nn.BatchNorm2d(num_features=mid_channels)
        self.activ = nn.ReLU(inplace=True)
        self.c_shuffle = ChannelShuffle(channels=out_channels, groups=2)
        self._pcv_runners = None

    def _runners(self):
        if self._pcv_runners is None:
            r = dict(c1=engine.ConvRunner(self.compress_conv1, self.compress_bn1),
                     d2=engine.ConvRunner(self.dw_conv2, self.dw_bn2),
                     e3=engine.ConvRunner(self.expand_conv3, self.expand_bn3))
            if self.downsample:
                r["d4"] = engine.ConvRunner(self.dw_conv4, self.dw_bn4)
                r["e5"] = engine.ConvRunner(self.expand_conv5, self.expand_bn5)
            self._pcv_runners = r
        return self._pcv_runners

    def _run(self, a):
        r = self._runners()
        relu = engine.act_code(self.activ)
        if self.downsample:
            y1 = r["e5"].run(r["d4"].run(a), act=relu)
            x2 = a
        else:
            y1 = a                                             # its first mid_channels channels are read in place
            x2 = engine.channel_slice(a, self.mid_channels, self.mid_channels)
        y2 = r["c1"].run(x2, act=relu)
        y2 = r["d2"].run(y2)
        if self.use_residual and not self.downsample:
            if self.use_se:
                raise NotImplementedError("ShuffleUnit with SE and residual")
            y2 = r["e3"].run(y2, act=relu, residual=x2)        # relu(bn(conv)) + x2, as the reference orders it
        else:
            y2 = r["e3"].run(y2, act=relu)
            if self.use_se:
                y2 = self.se(y2)
        return engine.cat_shuffle2(y1, y2, self.mid_channels)

    def forward(self, x):
        return engine.boundary(self, x, self._run)


class ShuffleInitBlock(nn.Module):
    """3x3/2 conv block + MaxPool2d(3, 2, padding=0, ceil_mode=True) (reference shufflenetv2.py:94-120)."""
    def __init__(self, in_channels, out_channels):
        super(ShuffleInitBlock, self).__init__()
        self.conv = conv3x3_block(in_channels=in_channels, out_channels=out_channels, stride=2)
        self.pool = MaxPool2dNHWC(kernel_size=3, stride=2, padding=0, ceil_mode=True)

    def forward(self, x):
        return engine.boundary(self, x, lambda a: self.pool(self.conv(a)), stem=True)


class ShuffleNetV2(nn.Module):
    def __init__(self, channels, init_block_channels, final_block_channels, use_se=False, use_residual=False, in_channels=3,
                 in_size=(224, 224), num_classes=1000):
        super(ShuffleNetV2, self).__init__()
        self.in_size = in_size
        self.num_classes = num_classes
        self.features = nn.Sequential()
        self.features.add_module("init_block", ShuffleInitBlock(in_channels=in_channels, out_channels=init_block_channels))
        in_channels = init_block_channels
        for i, channels_per_stage in enumerate(channels):
            stage = nn.Sequential()
            for j, out_channels in enumerate(channels_per_stage):
                stage.add_module("unit{}".format(j + 1), ShuffleUnit(in_channels=in_channels, out_channels=out_channels,
                                                                    downsample=(j == 0), use_se=use_se,
                                                                    use_residual=use_residual))
                in_channels = out_channels
            self.features.add_module("stage{}".format(i + 1), stage)
        self.features.add_module("final_block", conv1x1_block(in_channels=in_channels, out_channels=final_block_channels))
        in_channels = final_block_channels
        self.features.add_module("final_pool", AvgPool2dNHWC(kernel_size=7, stride=1, fp32_out=True))
        self.output = LinearHead(in_features=in_channels, out_features=num_classes)
        init_conv_params(self)

    def forward(self, x):
        return run_net(self, x, self.output)


def get_shufflenetv2(width_scale, model_name=None, pretrained=False, root=DEFAULT_ROOT, **kwargs):
    init_block_channels = 24
    final_block_channels = 1024
    layers = [4, 8, 4]
    channels_per_layers = [116, 232, 464]
    channels = [[ci] * li for (ci, li) in zip(channels_per_layers, layers)]
    if width_scale != 1.0:
        channels = [[int(cij * width_scale) for cij in ci] for ci in channels]
        if width_scale > 1.5:
            final_block_channels = int(final_block_channels * width_scale)
    net = ShuffleNetV2(channels=channels, init_block_channels=init_block_channels, final_block_channels=final_block_channels,
                       **kwargs)
    return maybe_load_pretrained(net, model_name, pretrained, root)


def shufflenetv2_wd2(**kwargs):
    return get_shufflenetv2(width_scale=(12.0 / 29.0), model_name="shufflenetv2_wd2", **kwargs)


def shufflenetv2_w1(**kwargs):
    return get_shufflenetv2(width_scale=1.0, model_name="shufflenetv2_w1", **kwargs)


def shufflenetv2_w3d2(**kwargs):
    return get_shufflenetv2(width_scale=(44.0 / 29.0), model_name="shufflenetv2_w3d2", **kwargs)


def shufflenetv2_w2(**kwargs):
    return get_shufflenetv2(width_scale=(61.0 / 29.0), model_name="shufflenetv2_w2", **kwargs)
