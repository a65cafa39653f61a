"""
    VGG / BN-VGG for ImageNet-1K on the MI355X hot path (reference pytorchcv/models/vgg.py:17-330): 3x3 convolutions with
    bias (+ BN) + ReLU as single fused launches, 2x2 max pools, and the three dense layers of the classifier. The reference
    flattens the 512x7x7 map in NCHW order before `fc1`; on the NHWC path `fc1` is the same weight matrix viewed as a 7x7
    valid convolution [4096, 512, 7, 7], so the flatten never materialises and the state_dict layout is unchanged.
"""

__all__ = ['VGG', 'get_vgg', 'vgg11', 'vgg13', 'vgg16', 'vgg19', 'bn_vgg11', 'bn_vgg13', 'bn_vgg16', 'bn_vgg19', 'bn_vgg11b',
           'bn_vgg13b', 'bn_vgg16b', 'bn_vgg19b']

import torch.nn as nn
from .. import engine
from .common.norm import lambda_batchnorm2d
from .common.conv import conv3x3_block
from ._tail import MaxPool2dNHWC, run_net, maybe_load_pretrained, init_conv_params, DEFAULT_ROOT


class _DenseAsConv(object):
    """Duck-typed view of an nn.Linear over a flattened [C, k, k] map as a k x k valid nn.Conv2d (weights shared)."""
    def __init__(self, lin, k):
        self._lin, self._k = lin, k
        self.in_channels, self.out_channels = lin.in_features // (k * k), lin.out_features
        self.kernel_size, self.stride, self.padding, self.dilation, self.groups = (k, k), (1, 1), (0, 0), (1, 1), 1
        self.padding_mode = "zeros"

    @property
    def weight(self):
        return self._lin.weight.view(self.out_channels, self.in_channels, self._k, self._k)

    @property
    def bias(self):
        return self._lin.bias


def _dense(lin, a, act, out_fp32=False):
    """`lin` applied to the NHWC handle `a` whose whole map is one sample's feature vector (NCHW-flatten order)."""
    if getattr(lin, "_pcv_runner", None) is None:
        if a.H != a.W or lin.in_features != a.C * a.H * a.W:
            raise RuntimeError("dense layer expects {} features, the map has {}x{}x{} (input size must match in_size)".format(
                lin.in_features, a.C, a.H, a.W))
        lin._pcv_runner = engine.ConvRunner(_DenseAsConv(lin, a.H), None)
    return lin._pcv_runner.run(a, act=act, out_fp32=out_fp32)


class VGGDense(nn.Module):
    """Linear + ReLU (+ Dropout, identity at inference) as one fused GEMM launch (reference vgg.py:17-42)."""
    def __init__(self, in_channels, out_channels):
        super(VGGDense, self).__init__()
        self.fc = nn.Linear(in_features=in_channels, out_features=out_channels)
        self.activ = nn.ReLU(inplace=True)
        self.dropout = nn.Dropout(p=0.5)

    def forward(self, x):
        return _dense(self.fc, x, engine.act_code(self.activ))


class VGGOutputBlock(nn.Module):
    """fc1 -> fc2 -> fc3 (reference vgg.py:45-77); fp32 logits [N, classes]."""
    def __init__(self, in_channels, classes):
        super(VGGOutputBlock, self).__init__()
        mid_channels = 4096
        self.fc1 = VGGDense(in_channels=in_channels, out_channels=mid_channels)
        self.fc2 = VGGDense(in_channels=mid_channels, out_channels=mid_channels)
        self.fc3 = nn.Linear(in_features=mid_channels, out_features=classes)

    def forward(self, x):
        if not isinstance(x, engine.NHWC):
            raise TypeError("VGGOutputBlock expects the NHWC handle of the last stage")
        x = self.fc2(self.fc1(x))
        y = _dense(self.fc3, x, 0, out_fp32=True)
        return y.t.view(y.N, -1)[:, :self.fc3.out_features]


class VGG(nn.Module):
    def __init__(self, channels, bias=True, use_bn=False, in_channels=3, in_size=(224, 224), num_classes=1000):
        super(VGG, self).__init__()
        self.in_size = in_size
        self.num_classes = num_classes
        normalization = lambda_batchnorm2d() if use_bn else None
        self.features = nn.Sequential()
        for i, channels_per_stage in enumerate(channels):
            stage = nn.Sequential()
            for j, out_channels in enumerate(channels_per_stage):
                stage.add_module("unit{}".format(j + 1), conv3x3_block(in_channels=in_channels, out_channels=out_channels,
                                                                       bias=bias, normalization=normalization))
                in_channels = out_channels
            stage.add_module("pool{}".format(i + 1), MaxPool2dNHWC(kernel_size=2, stride=2, padding=0))
            self.features.add_module("stage{}".format(i + 1), stage)
        self.output = VGGOutputBlock(in_channels=(in_channels * 7 * 7), classes=num_classes)
        init_conv_params(self)

    def forward(self, x):
        # the first convolution has stride 1: the 3-channel image goes in as a plain 8-channel-padded NHWC map
        return run_net(self, x, self.output, stem=False)


def get_vgg(blocks, bias=True, use_bn=False, model_name=None, pretrained=False, root=DEFAULT_ROOT, **kwargs):
    layers = {11: [1, 1, 2, 2, 2], 13: [2, 2, 2, 2, 2], 16: [2, 2, 3, 3, 3], 19: [2, 2, 4, 4, 4]}.get(blocks)
    if layers is None:
        raise ValueError("Unsupported VGG with number of blocks: {}".format(blocks))
    channels = [[ci] * li for (ci, li) in zip([64, 128, 256, 512, 512], layers)]
    net = VGG(channels=channels, bias=bias, use_bn=use_bn, **kwargs)
    return maybe_load_pretrained(net, model_name, pretrained, root)


def vgg11(**kwargs):
    return get_vgg(blocks=11, model_name="vgg11", **kwargs)


def vgg13(**kwargs):
    return get_vgg(blocks=13, model_name="vgg13", **kwargs)


def vgg16(**kwargs):
    return get_vgg(blocks=16, model_name="vgg16", **kwargs)


def vgg19(**kwargs):
    return get_vgg(blocks=19, model_name="vgg19", **kwargs)


def bn_vgg11(**kwargs):
    return get_vgg(blocks=11, bias=False, use_bn=True, model_name="bn_vgg11", **kwargs)


def bn_vgg13(**kwargs):
    return get_vgg(blocks=13, bias=False, use_bn=True, model_name="bn_vgg13", **kwargs)


def bn_vgg16(**kwargs):
    return get_vgg(blocks=16, bias=False, use_bn=True, model_name="bn_vgg16", **kwargs)


def bn_vgg19(**kwargs):
    return get_vgg(blocks=19, bias=False, use_bn=True, model_name="bn_vgg19", **kwargs)


def bn_vgg11b(**kwargs):
    return get_vgg(blocks=11, bias=True, use_bn=True, model_name="bn_vgg11b", **kwargs)


def bn_vgg13b(**kwargs):
    return get_vgg(blocks=13, bias=True, use_bn=True, model_name="bn_vgg13b", **kwargs)


def bn_vgg16b(**kwargs):
    return get_vgg(blocks=16, bias=True, use_bn=True, model_name="bn_vgg16b", **kwargs)


def bn_vgg19b(**kwargs):
    return get_vgg(blocks=19, bias=True, use_bn=True, model_name="bn_vgg19b", **kwargs)
