"""
    DenseNet for ImageNet-1K on the MI355X hot path (reference pytorchcv/models/densenet.py:16-270). A stage owns ONE buffer
    with the stage's final channel count; every DenseUnit reads the leading channels of it (BatchNorm + ReLU over a channel
    prefix, pcv_bn_act with a channel pitch) and its 3x3 convolution writes the 32 new channels straight into the next
    channel slice (`y_cpitch` of the fused convolution): `torch.cat((identity, x), dim=1)` never copies. The BN + ReLU
    between the unit's two convolutions is the 1x1's epilogue.
"""

__all__ = ['DenseNet', 'densenet121', 'densenet161', 'densenet169', 'densenet201', 'DenseUnit', 'TransitionBlock',
           'DenseStage', 'get_densenet']

import torch
import torch.nn as nn
from .common.conv import pre_conv1x1_block, pre_conv3x3_block
from .preresnet import PreResInitBlock, PreResActivation
from ._tail import AvgPool2dNHWC, LinearHead, run_net, maybe_load_pretrained, init_conv_params, DEFAULT_ROOT
from .. import engine


class DenseUnit(nn.Module):
    """BN-ReLU-1x1 -> BN-ReLU-3x3, output concatenated behind the input (reference densenet.py:16-59)."""
    def __init__(self, in_channels, out_channels, dropout_rate):
        super(DenseUnit, self).__init__()
        self.use_dropout = (dropout_rate != 0.0)
        bn_size = 4
        self.in_channels = in_channels
        self.inc_channels = out_channels - in_channels
        mid_channels = self.inc_channels * bn_size
        self.conv1 = pre_conv1x1_block(in_channels=in_channels, out_channels=mid_channels)
        self.conv2 = pre_conv3x3_block(in_channels=mid_channels, out_channels=self.inc_channels)
        if self.use_dropout:
            self.dropout = nn.Dropout(p=dropout_rate)      # identity at inference: never launched

    def run_into(self, buf):
        """`buf`: the stage buffer [N, H, W, Ctot] whose first `in_channels` channels are this unit's input; the unit's new
        channels go to [in_channels, in_channels + inc_channels)."""
        n, h, w, ctot = buf.shape
        prefix = engine.NHWC(buf, n, h, w, self.in_channels, cpitch=ctot)
        pre = self.conv1.preact(prefix)
        mid = self.conv1.conv_then(pre, next_block=self.conv2)
        self.conv2.conv_then(mid, out=(buf, self.in_channels))

    def _run(self, a):
        buf = torch.empty((a.N, a.H, a.W, self.in_channels + self.inc_channels), dtype=a.dtype, device=a.device)
        buf[:, :, :, :self.in_channels].copy_(a.t)
        self.run_into(buf)
        return engine.NHWC(buf, a.N, a.H, a.W, buf.shape[3])

    def forward(self, x):
        if self.training and self.use_dropout:
            raise RuntimeError("DenseUnit: the MI355X path is inference only (call net.eval())")
        return engine.boundary(self, x, self._run)


class TransitionBlock(nn.Module):
    """BN-ReLU-1x1 + 2x2 average pool (reference densenet.py:62-91)."""
    def __init__(self, in_channels, out_channels):
        super(TransitionBlock, self).__init__()
        self.conv = pre_conv1x1_block(in_channels=in_channels, out_channels=out_channels)
        self.pool = AvgPool2dNHWC(kernel_size=2, stride=2)

    def forward(self, x):
        return engine.boundary(self, x, lambda a: self.pool(self.conv(a)))


class DenseStage(nn.Sequential):
    """[transition +] dense units (a plain nn.Sequential in the reference, densenet.py:120-135: same children, same
    state_dict); on the hot path the units share one concatenation buffer."""
    def forward(self, x):
        if not isinstance(x, engine.NHWC):
            return super(DenseStage, self).forward(x)
        units = []
        for child in self.children():
            if isinstance(child, DenseUnit):
                units.append(child)
            else:
                if units:
                    raise RuntimeError("DenseStage expects the transition block before the dense units")
                x = child(x)
        if not units:
            return x
        ctot = units[-1].in_channels + units[-1].inc_channels
        buf = torch.empty((x.N, x.H, x.W, ctot), dtype=x.dtype, device=x.device)
        buf[:, :, :, :x.C].copy_(x.t)                 # the stage input becomes the first channel slice
        for unit in units:
            unit.run_into(buf)
        return engine.NHWC(buf, x.N, x.H, x.W, ctot)


class DenseNet(nn.Module):
    def __init__(self, channels, init_block_channels, dropout_rate=0.0, in_channels=3, in_size=(224, 224), num_classes=1000):
        super(DenseNet, self).__init__()
        self.in_size = in_size
        self.num_classes = num_classes
        self.features = nn.Sequential()
        self.features.add_module("init_block", PreResInitBlock(in_channels=in_channels, out_channels=init_block_channels))
        in_channels = init_block_channels
        for i, channels_per_stage in enumerate(channels):
            stage = DenseStage()
            if i != 0:
                stage.add_module("trans{}".format(i + 1), TransitionBlock(in_channels=in_channels,
                                                                          out_channels=(in_channels // 2)))
                in_channels = in_channels // 2
            for j, out_channels in enumerate(channels_per_stage):
                stage.add_module("unit{}".format(j + 1), DenseUnit(in_channels=in_channels, out_channels=out_channels,
                                                                  dropout_rate=dropout_rate))
                in_channels = out_channels
            self.features.add_module("stage{}".format(i + 1), stage)
        self.features.add_module("post_activ", PreResActivation(in_channels=in_channels))
        self.features.add_module("final_pool", AvgPool2dNHWC(kernel_size=7, stride=1, fp32_out=True))
        self.output = LinearHead(in_features=in_channels, out_features=num_classes)
        init_conv_params(self)

    def forward(self, x):
        return run_net(self, x, self.output)


# blocks -> (init_block_channels, growth_rate, layers), reference densenet.py:204-221
_VERSIONS = {121: (64, 32, [6, 12, 24, 16]), 161: (96, 48, [6, 12, 36, 24]), 169: (64, 32, [6, 12, 32, 32]),
             201: (64, 32, [6, 12, 48, 32])}


def get_densenet(blocks, model_name=None, pretrained=False, root=DEFAULT_ROOT, **kwargs):
    if blocks not in _VERSIONS:
        raise ValueError("Unsupported DenseNet version with number of layers {}".format(blocks))
    init_block_channels, growth_rate, layers = _VERSIONS[blocks]
    channels = []
    width = init_block_channels * 2                     # the first stage has no transition: it starts from 2x / 2
    for count in layers:
        width //= 2
        stage = []
        for _ in range(count):
            width += growth_rate
            stage.append(width)
        channels.append(stage)
    net = DenseNet(channels=channels, init_block_channels=init_block_channels, **kwargs)
    return maybe_load_pretrained(net, model_name, pretrained, root)


def densenet121(**kwargs):
    return get_densenet(blocks=121, model_name="densenet121", **kwargs)


def densenet161(**kwargs):
    return get_densenet(blocks=161, model_name="densenet161", **kwargs)


def densenet169(**kwargs):
    return get_densenet(blocks=169, model_name="densenet169", **kwargs)


def densenet201(**kwargs):
    return get_densenet(blocks=201, model_name="densenet201", **kwargs)
