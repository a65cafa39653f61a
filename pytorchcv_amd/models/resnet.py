"""
    ResNet for ImageNet-1K on the MI355X hot path. Module tree, attribute names and factory signatures follow the reference
    (pytorchcv/models/resnet.py:19-442) so its state_dicts load strictly; every unit is executed as fused launches:
    a bottleneck unit = 3 (or 4 with `identity_conv`) kernels, the residual add + ReLU living in the last one's epilogue.
"""

__all__ = ['ResNet', 'resnet10', 'resnet12', 'resnet14', 'resnetbc14b', 'resnet16', 'resnet18', 'resnet26', 'resnetbc26b',
           'resnet34', 'resnetbc38b', 'resnet50', 'resnet50b', 'resnet101', 'resnet101b', 'resnet152', 'resnet152b',
           'ResBlock', 'ResBottleneck', 'ResUnit', 'ResStage', 'ResInitBlock', 'get_resnet']

import torch.nn as nn
from .common.activ import lambda_relu
from .common.norm import lambda_batchnorm2d
from .common.conv import conv1x1_block, conv3x3_block, conv7x7_block, conv_block_pair, conv_block_maxpool
from ._tail import MaxPool2dNHWC, AvgPool2dNHWC, LinearHead, run_net, maybe_load_pretrained, init_conv_params, DEFAULT_ROOT
from .. import engine


class ResBlock(nn.Module):
    """Two 3x3 blocks (reference resnet.py:19-66); `residual`/`post_act` ride on the second one."""
    def __init__(self, in_channels, out_channels, stride, bias=False, normalization=lambda_batchnorm2d(),
                 activation=lambda_relu(), final_activation=None):
        super(ResBlock, self).__init__()
        self.conv1 = conv3x3_block(in_channels=in_channels, out_channels=out_channels, stride=stride, bias=bias,
                                   normalization=normalization, activation=activation)
        self.conv2 = conv3x3_block(in_channels=out_channels, out_channels=out_channels, bias=bias,
                                   normalization=normalization, activation=final_activation)

    def forward(self, x, residual=None, post_act=None):
        return engine.boundary(self, x, lambda a: self.conv2(self.conv1(a), residual=residual, post_act=post_act))


class ResBottleneck(nn.Module):
    """1x1 -> 3x3 -> 1x1 (reference resnet.py:69-140); the stride sits on conv1 when `conv1_stride`."""
    def __init__(self, in_channels, out_channels, stride, padding=1, dilation=1, bias=False,
                 normalization=lambda_batchnorm2d(), conv1_stride=False, bottleneck_factor=4, activation=lambda_relu(),
                 final_activation=None):
        super(ResBottleneck, self).__init__()
        mid_channels = out_channels // bottleneck_factor
        self.conv1 = conv1x1_block(in_channels=in_channels, out_channels=mid_channels, stride=(stride if conv1_stride else 1),
                                   bias=bias, normalization=normalization, activation=activation)
        self.conv2 = conv3x3_block(in_channels=mid_channels, out_channels=mid_channels,
                                   stride=(1 if conv1_stride else stride), padding=padding, dilation=dilation, bias=bias,
                                   normalization=normalization, activation=activation)
        self.conv3 = conv1x1_block(in_channels=mid_channels, out_channels=out_channels, bias=bias,
                                   normalization=normalization, activation=final_activation)

    def forward(self, x, residual=None, post_act=None):
        return engine.boundary(self, x, lambda a: self.conv3(self.conv2(self.conv1(a)), residual=residual, post_act=post_act))


class ResUnit(nn.Module):
    """relu(body(x) + identity) (reference resnet.py:143-229)."""
    def __init__(self, in_channels, out_channels, stride=1, padding=1, dilation=1, bias=False,
                 normalization=lambda_batchnorm2d(), bottleneck=True, conv1_stride=False, activation=lambda_relu(),
                 final_body_activation=None, final_activation=lambda_relu()):
        super(ResUnit, self).__init__()
        self.resize_identity = (in_channels != out_channels) or (stride != 1)
        if bottleneck:
            self.body = ResBottleneck(in_channels=in_channels, out_channels=out_channels, stride=stride, padding=padding,
                                      dilation=dilation, bias=bias, normalization=normalization, conv1_stride=conv1_stride,
                                      activation=activation, final_activation=final_body_activation)
        else:
            self.body = ResBlock(in_channels=in_channels, out_channels=out_channels, stride=stride, bias=bias,
                                 normalization=normalization, activation=activation, final_activation=final_body_activation)
        if self.resize_identity:
            self.identity_conv = conv1x1_block(in_channels=in_channels, out_channels=out_channels, stride=stride, bias=bias,
                                               normalization=normalization, activation=None)
        self.activ = final_activation()

    def _run(self, a):
        identity = self.identity_conv(a) if self.resize_identity else a
        return self.body(a, residual=identity, post_act=self.activ)

    def run_chained(self, a, conv1_out=None, next_unit=None):
        """Stage-level execution of a bottleneck unit: `conv1_out` is this unit's first convolution when the previous unit
        already produced it; with `next_unit`, the last convolution (+ skip add + ReLU) and the next unit's first
        convolution go out as one launch whenever the pair is covered. Returns (unit output, next unit's conv1 output
        or None)."""
        body = self.body
        y = body.conv2(conv1_out if conv1_out is not None else body.conv1(a))
        if next_unit is not None and self.resize_identity:
            # the skip convolution recomputed inside the fused pair (no launch, no 4x wider tensor written and re-read)
            pair = conv_block_pair(body.conv3, y, None, self.activ, next_unit.body.conv1, id_block=self.identity_conv, x0=a)
            if pair is not None:
                return pair
        identity = self.identity_conv(a) if self.resize_identity else a
        if next_unit is not None:
            pair = conv_block_pair(body.conv3, y, identity, self.activ, next_unit.body.conv1)
            if pair is not None:
                return pair
        return body.conv3(y, residual=identity, post_act=self.activ), None

    def forward(self, x):
        return engine.boundary(self, x, self._run)


class ResStage(nn.Sequential):
    """A stage of units (plain nn.Sequential in the reference, resnet.py:297-313; same child names, same state_dict). On
    the hot path consecutive bottleneck units are chained so that a unit's last 1x1 convolution and the next unit's
    first one can share a launch."""
    def forward(self, x):
        units = list(self.children())
        chainable = all((isinstance(u, ResUnit) and isinstance(u.body, ResBottleneck)) or getattr(u, "pcv_chainable", False) for u in units)
        if not isinstance(x, engine.NHWC) or not chainable:
            return super(ResStage, self).forward(x)
        conv1_out = None
        for i, unit in enumerate(units):
            x, conv1_out = unit.run_chained(x, conv1_out, units[i + 1] if i + 1 < len(units) else None)
        return x


class ResInitBlock(nn.Module):
    """7x7/2 stem + 3x3/2 max-pool (reference resnet.py:232-263)."""
    def __init__(self, in_channels, out_channels, normalization=lambda_batchnorm2d()):
        super(ResInitBlock, self).__init__()
        self.conv = conv7x7_block(in_channels=in_channels, out_channels=out_channels, stride=2, normalization=normalization)
        self.pool = MaxPool2dNHWC(kernel_size=3, stride=2, padding=1)

    def forward(self, x):
        return engine.boundary(self, x, lambda a: conv_block_maxpool(self.conv, a, self.pool), stem=True)


class ResNet(nn.Module):
    """`features` (init_block, stage1..4 of unit1..n, final_pool) + `output` Linear (reference resnet.py:266-337)."""
    def __init__(self, channels, init_block_channels, bottleneck, conv1_stride, in_channels=3, in_size=(224, 224),
                 num_classes=1000):
        super(ResNet, self).__init__()
        self.in_size = in_size
        self.num_classes = num_classes
        self.features = nn.Sequential()
        self.features.add_module("init_block", ResInitBlock(in_channels=in_channels, out_channels=init_block_channels))
        in_channels = init_block_channels
        for i, channels_per_stage in enumerate(channels):
            stage = ResStage()
            for j, out_channels in enumerate(channels_per_stage):
                stride = 2 if (j == 0) and (i != 0) else 1
                stage.add_module("unit{}".format(j + 1), ResUnit(in_channels=in_channels, out_channels=out_channels,
                                                                 stride=stride, bottleneck=bottleneck,
                                                                 conv1_stride=conv1_stride))
                in_channels = out_channels
            self.features.add_module("stage{}".format(i + 1), stage)
        self.features.add_module("final_pool", AvgPool2dNHWC(kernel_size=7, stride=1, fp32_out=True))
        self.output = LinearHead(in_features=in_channels, out_features=num_classes)
        init_conv_params(self)

    def forward(self, x):
        return run_net(self, x, self.output)


_LAYERS = {10: [1, 1, 1, 1], 12: [2, 1, 1, 1], 16: [2, 2, 2, 1], 18: [2, 2, 2, 2], 34: [3, 4, 6, 3], 50: [3, 4, 6, 3],
           101: [3, 4, 23, 3], 152: [3, 8, 36, 3], 200: [3, 24, 36, 3]}


def resnet_layers(blocks, bottleneck, what="ResNet"):
    """Units per stage for a depth (reference resnet.py:373-411)."""
    if blocks == 14:
        layers = [1, 1, 1, 1] if bottleneck else [2, 2, 1, 1]
    elif blocks == 26:
        layers = [2, 2, 2, 2] if bottleneck else [3, 3, 3, 3]
    elif blocks == 38 and bottleneck:
        layers = [3, 3, 3, 3]
    elif blocks in _LAYERS:
        layers = _LAYERS[blocks]
    else:
        raise ValueError("Unsupported {} with number of blocks: {}".format(what, blocks))
    assert (sum(layers) * (3 if bottleneck else 2) + 2 == blocks)
    return layers


def get_resnet(blocks, bottleneck=None, conv1_stride=True, width_scale=1.0, model_name=None, pretrained=False,
               root=DEFAULT_ROOT, **kwargs):
    if bottleneck is None:
        bottleneck = (blocks >= 50)
    layers = resnet_layers(blocks, bottleneck)
    init_block_channels = 64
    widths = [64, 128, 256, 512]
    if bottleneck:
        widths = [w * 4 for w in widths]
    channels = [[w] * n for (w, n) in zip(widths, layers)]
    if width_scale != 1.0:
        last = (len(channels) - 1, len(channels[-1]) - 1)
        channels = [[int(c * width_scale) if (i, j) != last else c for j, c in enumerate(ci)] for i, ci in enumerate(channels)]
        init_block_channels = int(init_block_channels * width_scale)
    net = ResNet(channels=channels, init_block_channels=init_block_channels, bottleneck=bottleneck, conv1_stride=conv1_stride,
                 **kwargs)
    return maybe_load_pretrained(net, model_name, pretrained, root)


def resnet10(**kwargs):
    return get_resnet(blocks=10, model_name="resnet10", **kwargs)


def resnet12(**kwargs):
    return get_resnet(blocks=12, model_name="resnet12", **kwargs)


def resnet14(**kwargs):
    return get_resnet(blocks=14, model_name="resnet14", **kwargs)


def resnetbc14b(**kwargs):
    return get_resnet(blocks=14, bottleneck=True, conv1_stride=False, model_name="resnetbc14b", **kwargs)


def resnet16(**kwargs):
    return get_resnet(blocks=16, model_name="resnet16", **kwargs)


def resnet18(**kwargs):
    return get_resnet(blocks=18, model_name="resnet18", **kwargs)


def resnet26(**kwargs):
    return get_resnet(blocks=26, bottleneck=False, model_name="resnet26", **kwargs)


def resnetbc26b(**kwargs):
    return get_resnet(blocks=26, bottleneck=True, conv1_stride=False, model_name="resnetbc26b", **kwargs)


def resnet34(**kwargs):
    return get_resnet(blocks=34, model_name="resnet34", **kwargs)


def resnetbc38b(**kwargs):
    return get_resnet(blocks=38, bottleneck=True, conv1_stride=False, model_name="resnetbc38b", **kwargs)


def resnet50(**kwargs):
    return get_resnet(blocks=50, model_name="resnet50", **kwargs)


def resnet50b(**kwargs):
    return get_resnet(blocks=50, conv1_stride=False, model_name="resnet50b", **kwargs)


def resnet101(**kwargs):
    return get_resnet(blocks=101, model_name="resnet101", **kwargs)


def resnet101b(**kwargs):
    return get_resnet(blocks=101, conv1_stride=False, model_name="resnet101b", **kwargs)


def resnet152(**kwargs):
    return get_resnet(blocks=152, model_name="resnet152", **kwargs)


def resnet152b(**kwargs):
    return get_resnet(blocks=152, conv1_stride=False, model_name="resnet152b", **kwargs)
