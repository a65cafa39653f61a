"""
    SE-PreResNet for ImageNet-1K on the MI355X hot path (reference pytorchcv/models/sepreresnet.py:17-560): PreResNet bodies
    with an SEBlock between the body and the skip add: in the bottleneck nets it runs inside the body's last 1x1 convolution
    (SEBlock.run_behind / PreConvBlock.conv_then(se=...)), in the basic-block nets the channel scale and the add are one pass (pcv_se_scale).
"""

__all__ = ['SEPreResNet', 'sepreresnet10', 'sepreresnet12', 'sepreresnet14', 'sepreresnet16', 'sepreresnet18',
           'sepreresnet26', 'sepreresnetbc26b', 'sepreresnet34', 'sepreresnetbc38b', 'sepreresnet50', 'sepreresnet50b',
           'sepreresnet101', 'sepreresnet101b', 'sepreresnet152', 'sepreresnet152b', 'sepreresnet200',
           'sepreresnet200b', 'SEPreResUnit', 'get_sepreresnet']

import torch.nn as nn
from .common.conv import conv1x1
from .common.att import SEBlock
from .resnet import resnet_layers
from .preresnet import PreResBlock, PreResBottleneck, PreResInitBlock, PreResActivation, _chain_forward
from ._tail import AvgPool2dNHWC, LinearHead, run_net, maybe_load_pretrained, init_conv_params, DEFAULT_ROOT
from .. import engine


class SEPreResUnit(nn.Module):
    """reference sepreresnet.py:17-71: body -> SE -> + identity (identity_conv reads the pre-activated input)."""
    def __init__(self, in_channels, out_channels, stride, bottleneck, conv1_stride):
        super(SEPreResUnit, self).__init__()
        self.resize_identity = (in_channels != out_channels) or (stride != 1)
        if bottleneck:
            self.body = PreResBottleneck(in_channels=in_channels, out_channels=out_channels, stride=stride,
                                         conv1_stride=conv1_stride)
        else:
            self.body = PreResBlock(in_channels=in_channels, out_channels=out_channels, stride=stride)
        self.se = SEBlock(channels=out_channels)
        if self.resize_identity:
            self.identity_conv = conv1x1(in_channels=in_channels, out_channels=out_channels, stride=stride)

    def _run(self, a):
        blocks = self.body.chain()
        pre = blocks[0].preact(a)
        identity = self.identity_conv(pre) if self.resize_identity else a
        return _chain_forward(blocks, pre, residual=identity, se=self.se)

    def forward(self, x):
        return engine.boundary(self, x, self._run)


class SEPreResNet(nn.Module):
    def __init__(self, channels, init_block_channels, bottleneck, conv1_stride, in_channels=3, in_size=(224, 224),
                 num_classes=1000):
        super(SEPreResNet, self).__init__()
        self.in_size = in_size
        self.num_classes = num_classes
        self.features = nn.Sequential()
        self.features.add_module("init_block", PreResInitBlock(in_channels=in_channels, out_channels=init_block_channels))
        in_channels = init_block_channels
        for i, channels_per_stage in enumerate(channels):
            stage = nn.Sequential()
            for j, out_channels in enumerate(channels_per_stage):
                stride = 1 if (i == 0) or (j != 0) else 2
                stage.add_module("unit{}".format(j + 1), SEPreResUnit(in_channels=in_channels, out_channels=out_channels,
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


def get_sepreresnet(blocks, bottleneck=None, conv1_stride=True, model_name=None, pretrained=False, root=DEFAULT_ROOT, **kwargs):
    if bottleneck is None:
        bottleneck = (blocks >= 50)
    layers = [3, 30, 48, 8] if blocks == 269 else resnet_layers(blocks, bottleneck, what="SE-PreResNet")
    assert (sum(layers) * (3 if bottleneck else 2) + 2 == blocks)
    channels_per_layers = [64, 128, 256, 512]
    if bottleneck:
        channels_per_layers = [ci * 4 for ci in channels_per_layers]
    channels = [[ci] * li for (ci, li) in zip(channels_per_layers, layers)]
    net = SEPreResNet(channels=channels, init_block_channels=64, bottleneck=bottleneck, conv1_stride=conv1_stride, **kwargs)
    return maybe_load_pretrained(net, model_name, pretrained, root)


# name -> get_sepreresnet arguments (reference sepreresnet.py:239-560)
_VARIANTS = {
    "sepreresnet10": dict(blocks=10), "sepreresnet12": dict(blocks=12), "sepreresnet14": dict(blocks=14),
    "sepreresnet16": dict(blocks=16), "sepreresnet18": dict(blocks=18), "sepreresnet26": dict(blocks=26, bottleneck=False),
    "sepreresnetbc26b": dict(blocks=26, bottleneck=True, conv1_stride=False), "sepreresnet34": dict(blocks=34),
    "sepreresnetbc38b": dict(blocks=38, bottleneck=True, conv1_stride=False), "sepreresnet50": dict(blocks=50),
    "sepreresnet50b": dict(blocks=50, conv1_stride=False), "sepreresnet101": dict(blocks=101),
    "sepreresnet101b": dict(blocks=101, conv1_stride=False), "sepreresnet152": dict(blocks=152),
    "sepreresnet152b": dict(blocks=152, conv1_stride=False), "sepreresnet200": dict(blocks=200),
    "sepreresnet200b": dict(blocks=200, conv1_stride=False),
}


def _variant(name, args):
    def factory(**kwargs):
        return get_sepreresnet(model_name=name, **dict(args, **kwargs))
    factory.__name__ = name
    factory.__doc__ = "SE-PreResNet variant `{}` (reference sepreresnet.py:239-560).".format(name)
    return factory


for _name, _args in _VARIANTS.items():
    globals()[_name] = _variant(_name, _args)
