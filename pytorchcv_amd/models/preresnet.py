"""
    PreResNet for ImageNet-1K on the MI355X hot path (reference pytorchcv/models/preresnet.py:19-330): pre-activation units.
    Inside a unit every BatchNorm+ReLU that follows a convolution rides in that convolution's epilogue (conv1 carries
    conv2's pre-activation, conv2 carries conv3's); only the unit's first pre-activation, whose input the skip path needs
    raw, is its own elementwise launch. The skip add is the epilogue of the unit's last convolution.
"""

__all__ = ['PreResNet', 'preresnet10', 'preresnet12', 'preresnet14', 'preresnetbc14b', 'preresnet16', 'preresnet18_wd4',
           'preresnet18_wd2', 'preresnet18_w3d4', 'preresnet18', 'preresnet26', 'preresnetbc26b', 'preresnet34',
           'preresnetbc38b', 'preresnet50', 'preresnet50b', 'preresnet101', 'preresnet101b', 'preresnet152',
           'preresnet152b', 'preresnet200', 'preresnet200b', 'preresnet269b', 'PreResBlock', 'PreResBottleneck',
           'PreResUnit', 'PreResInitBlock', 'PreResActivation', 'get_preresnet']

import torch.nn as nn
from .common.norm import lambda_batchnorm2d
from .common.conv import pre_conv1x1_block, pre_conv3x3_block, conv1x1
from .resnet import resnet_layers
from ._tail import MaxPool2dNHWC, AvgPool2dNHWC, LinearHead, run_net, maybe_load_pretrained, init_conv_params, DEFAULT_ROOT
from .. import engine


class PreResBlock(nn.Module):
    """Two pre-activated 3x3 convolutions (reference preresnet.py:19-61); returns (x, x_pre_activ)."""
    def __init__(self, in_channels, out_channels, stride, bias=False, normalization=lambda_batchnorm2d()):
        super(PreResBlock, self).__init__()
        self.conv1 = pre_conv3x3_block(in_channels=in_channels, out_channels=out_channels, stride=stride, bias=bias,
                                       normalization=normalization, return_preact=True)
        self.conv2 = pre_conv3x3_block(in_channels=out_channels, out_channels=out_channels, bias=bias,
                                       normalization=normalization)

    def chain(self):
        return [self.conv1, self.conv2]

    def forward(self, x):
        return _run_body(self, x)


class PreResBottleneck(nn.Module):
    """1x1 -> 3x3 -> 1x1 pre-activated bottleneck (reference preresnet.py:64-106)."""
    def __init__(self, in_channels, out_channels, stride, conv1_stride):
        super(PreResBottleneck, self).__init__()
        mid_channels = out_channels // 4
        self.conv1 = pre_conv1x1_block(in_channels=in_channels, out_channels=mid_channels,
                                       stride=(stride if conv1_stride else 1), return_preact=True)
        self.conv2 = pre_conv3x3_block(in_channels=mid_channels, out_channels=mid_channels,
                                       stride=(1 if conv1_stride else stride))
        self.conv3 = pre_conv1x1_block(in_channels=mid_channels, out_channels=out_channels)

    def chain(self):
        return [self.conv1, self.conv2, self.conv3]

    def forward(self, x):
        return _run_body(self, x)


def _chain_forward(blocks, a, residual=None, se=None):
    """`a` is already pre-activated for blocks[0]; block i's convolution carries block i+1's BN+activation. `se`: the SEBlock
    that follows the body (SE-PreResNet): it runs inside the last convolution when that is a 1x1 (PreConvBlock.conv_then)."""
    for i, blk in enumerate(blocks):
        last = (i + 1 == len(blocks))
        a = blk.conv_then(a, next_block=(None if last else blocks[i + 1]), residual=(residual if last else None),
                          se=(se if last else None))
    return a


def _run_body(body, x):
    """Body used on its own (drop-in block): (x, x_pre_activ), NCHW in -> NCHW out."""
    def run(a):
        blocks = body.chain()
        pre = blocks[0].preact(a)
        return _chain_forward(blocks, pre), pre
    return engine.boundary(body, x, run)               # (tensor entry: the module's resolved type + the fp16 range guard)


class PreResUnit(nn.Module):
    """Pre-activation residual unit (reference preresnet.py:109-164): the identity convolution, when present, reads the
    pre-activated input and has neither BN nor activation."""
    def __init__(self, in_channels, out_channels, stride, bias=False, normalization=lambda_batchnorm2d(), bottleneck=True,
                 conv1_stride=False):
        super(PreResUnit, self).__init__()
        self.resize_identity = (in_channels != out_channels) or (stride != 1)
        if bottleneck:
            self.body = PreResBottleneck(in_channels=in_channels, out_channels=out_channels, stride=stride,
                                         conv1_stride=conv1_stride)
        else:
            self.body = PreResBlock(in_channels=in_channels, out_channels=out_channels, stride=stride, bias=bias,
                                    normalization=normalization)
        if self.resize_identity:
            self.identity_conv = conv1x1(in_channels=in_channels, out_channels=out_channels, stride=stride, bias=bias)

    def _run(self, a):
        blocks = self.body.chain()
        pre = blocks[0].preact(a)
        identity = self.identity_conv(pre) if self.resize_identity else a
        return _chain_forward(blocks, pre, residual=identity)

    def forward(self, x):
        return engine.boundary(self, x, self._run)


class PreResInitBlock(nn.Module):
    """7x7/2 conv + BN + ReLU + 3x3/2 max-pool with the parameters directly on the block (reference preresnet.py:167-196)."""
    def __init__(self, in_channels, out_channels):
        super(PreResInitBlock, self).__init__()
        self.conv = nn.Conv2d(in_channels=in_channels, out_channels=out_channels, kernel_size=7, stride=2, padding=3,
                              bias=False)
        self.bn = nn.BatchNorm2d(num_features=out_channels)
        self.activ = nn.ReLU(inplace=True)
        self.pool = MaxPool2dNHWC(kernel_size=3, stride=2, padding=1)
        self._pcv_runner = None

    def _run(self, a):
        if self._pcv_runner is None:
            self._pcv_runner = engine.ConvRunner(self.conv, self.bn)
        y = self._pcv_runner.run_maxpool(a, engine.act_code(self.activ), self.pool.kernel_size, self.pool.stride, self.pool.padding,
                                         self.pool.ceil_mode)
        return y if y is not None else self.pool(self._pcv_runner.run(a, act=engine.act_code(self.activ)))

    def forward(self, x):
        return engine.boundary(self, x, self._run, stem=True)


class PreResActivation(nn.Module):
    """The BatchNorm + ReLU that closes the pre-activation trunk (reference preresnet.py:199-222)."""
    def __init__(self, in_channels):
        super(PreResActivation, self).__init__()
        self.bn = nn.BatchNorm2d(num_features=in_channels)
        self.activ = nn.ReLU(inplace=True)
        self._pcv_pre = None

    def _run(self, a):
        if self._pcv_pre is None:
            self._pcv_pre = engine.BnActRunner(self.bn)
        return self._pcv_pre.run(a, engine.act_code(self.activ))

    def forward(self, x):
        return engine.boundary(self, x, self._run)


class PreResNet(nn.Module):
    def __init__(self, channels, init_block_channels, bottleneck, conv1_stride, in_channels=3, in_size=(224, 224),
                 num_classes=1000):
        super(PreResNet, self).__init__()
        self.in_size = in_size
        self.num_classes = num_classes
        self.features = nn.Sequential()
        self.features.add_module("init_block", PreResInitBlock(in_channels=in_channels, out_channels=init_block_channels))
        in_channels = init_block_channels
        for i, channels_per_stage in enumerate(channels):
            stage = nn.Sequential()
            for j, out_channels in enumerate(channels_per_stage):
                stride = 1 if (i == 0) or (j != 0) else 2
                stage.add_module("unit{}".format(j + 1), PreResUnit(in_channels=in_channels, out_channels=out_channels,
                                                                   stride=stride, bottleneck=bottleneck,
                                                                   conv1_stride=conv1_stride))
                in_channels = out_channels
            self.features.add_module("stage{}".format(i + 1), stage)
        self.features.add_module("post_activ", PreResActivation(in_channels=in_channels))
        self.features.add_module("final_pool", AvgPool2dNHWC(kernel_size=7, stride=1, fp32_out=True))
        self.output = LinearHead(in_features=in_channels, out_features=num_classes)
        init_conv_params(self)

    def forward(self, x):
        return run_net(self, x, self.output)


def get_preresnet(blocks, bottleneck=None, conv1_stride=True, width_scale=1.0, model_name=None, pretrained=False,
                  root=DEFAULT_ROOT, **kwargs):
    """Depth table and channel plan of reference preresnet.py:320-369 (same table as ResNet plus 269)."""
    if bottleneck is None:
        bottleneck = (blocks >= 50)
    if blocks == 269:
        layers = [3, 30, 48, 8]
    else:
        layers = resnet_layers(blocks, bottleneck, what="PreResNet")
    assert (sum(layers) * (3 if bottleneck else 2) + 2 == blocks)
    init_block_channels = 64
    channels_per_layers = [64, 128, 256, 512]
    if bottleneck:
        channels_per_layers = [ci * 4 for ci in channels_per_layers]
    channels = [[ci] * li for (ci, li) in zip(channels_per_layers, layers)]
    if width_scale != 1.0:
        channels = [[int(cij * width_scale) if (i != len(channels) - 1) or (j != len(ci) - 1) else cij
                     for j, cij in enumerate(ci)] for i, ci in enumerate(channels)]
        init_block_channels = int(init_block_channels * width_scale)
    net = PreResNet(channels=channels, init_block_channels=init_block_channels, bottleneck=bottleneck,
                    conv1_stride=conv1_stride, **kwargs)
    return maybe_load_pretrained(net, model_name, pretrained, root)


# name -> get_preresnet arguments (reference preresnet.py:372-878)
_VARIANTS = {
    "preresnet10": dict(blocks=10), "preresnet12": dict(blocks=12), "preresnet14": dict(blocks=14),
    "preresnetbc14b": dict(blocks=14, bottleneck=True, conv1_stride=False), "preresnet16": dict(blocks=16),
    "preresnet18_wd4": dict(blocks=18, width_scale=0.25), "preresnet18_wd2": dict(blocks=18, width_scale=0.5),
    "preresnet18_w3d4": dict(blocks=18, width_scale=0.75), "preresnet18": dict(blocks=18),
    "preresnet26": dict(blocks=26, bottleneck=False), "preresnetbc26b": dict(blocks=26, bottleneck=True, conv1_stride=False),
    "preresnet34": dict(blocks=34), "preresnetbc38b": dict(blocks=38, bottleneck=True, conv1_stride=False),
    "preresnet50": dict(blocks=50), "preresnet50b": dict(blocks=50, conv1_stride=False),
    "preresnet101": dict(blocks=101), "preresnet101b": dict(blocks=101, conv1_stride=False),
    "preresnet152": dict(blocks=152), "preresnet152b": dict(blocks=152, conv1_stride=False),
    "preresnet200": dict(blocks=200), "preresnet200b": dict(blocks=200, conv1_stride=False),
    "preresnet269b": dict(blocks=269, conv1_stride=False),
}


def _variant(name, args):
    def factory(**kwargs):
        return get_preresnet(model_name=name, **dict(args, **kwargs))
    factory.__name__ = name
    factory.__doc__ = "PreResNet variant `{}` (reference preresnet.py:372-878).".format(name)
    return factory


for _name, _args in _VARIANTS.items():
    globals()[_name] = _variant(_name, _args)
