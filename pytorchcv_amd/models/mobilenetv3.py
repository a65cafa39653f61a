"""
    MobileNetV3 for ImageNet-1K on the MI355X hot path (reference pytorchcv/models/mobilenetv3.py:18-595): expand 1x1 (MFMA),
    depthwise 3x3/5x5 (direct kernel), SE with a hard-sigmoid gate, project 1x1 with the skip add in its epilogue; h-swish
    is an epilogue code, the classifier is two 1x1 GEMMs (h-swish fused into the first, bias + fp32 logits in the second).
"""

__all__ = ['MobileNetV3', 'mobilenetv3_small_w7d20', 'mobilenetv3_small_wd2', 'mobilenetv3_small_w3d4',
           'mobilenetv3_small_w1', 'mobilenetv3_small_w5d4', 'mobilenetv3_large_w7d20', 'mobilenetv3_large_wd2',
           'mobilenetv3_large_w3d4', 'mobilenetv3_large_w1', 'mobilenetv3_large_w5d4', 'MobileNetV3Unit',
           'MobileNetV3FinalBlock', 'MobileNetV3Classifier', 'get_mobilenetv3']

import torch.nn as nn
from .common.activ import lambda_relu, lambda_hswish, lambda_hsigmoid, HSwish
from .common.conv import conv1x1, conv1x1_block, conv3x3_block, dwconv3x3_block, dwconv5x5_block, mbconv_chain
from .common.att import SEBlock, round_channels
from ._tail import AvgPool2dNHWC, run_net, maybe_load_pretrained, init_conv_params, DEFAULT_ROOT
from .. import engine


class MobileNetV3Unit(nn.Module):
    """Inverted residual with optional SE between the depthwise and the projection (reference mobilenetv3.py:18-93)."""
    def __init__(self, in_channels, out_channels, exp_channels, stride, use_kernel3, activation, use_se):
        super(MobileNetV3Unit, self).__init__()
        assert (exp_channels >= out_channels)
        self.residual = (in_channels == out_channels) and (stride == 1)
        self.use_se = use_se
        self.use_exp_conv = exp_channels != out_channels
        mid_channels = exp_channels
        if self.use_exp_conv:
            self.exp_conv = conv1x1_block(in_channels=in_channels, out_channels=mid_channels, activation=activation)
        dw_block = dwconv3x3_block if use_kernel3 else dwconv5x5_block
        self.conv1 = dw_block(in_channels=mid_channels, out_channels=mid_channels, stride=stride, activation=activation)
        if self.use_se:
            self.se = SEBlock(channels=mid_channels, reduction=4, round_mid=True, out_activation=lambda_hsigmoid())
        self.conv2 = conv1x1_block(in_channels=mid_channels, out_channels=out_channels, activation=None)

    def _run(self, a):
        if not self.use_se:
            y = mbconv_chain(self.exp_conv if self.use_exp_conv else None, self.conv1, self.conv2, a,
                             residual=(a if self.residual else None))
            if y is not None:
                return y                                 # the whole unit was one launch (csrc/mbconv.hpp)
        y = self.exp_conv(a) if self.use_exp_conv else a
        y = self.conv1(y)
        if self.use_se:
            y = self.se(y)
        return self.conv2(y, residual=(a if self.residual else None))

    def forward(self, x):
        return engine.boundary(self, x, self._run)


class MobileNetV3FinalBlock(nn.Module):
    """1x1 h-swish block + optional SE (reference mobilenetv3.py:96-131)."""
    def __init__(self, in_channels, out_channels, use_se):
        super(MobileNetV3FinalBlock, self).__init__()
        self.use_se = use_se
        self.conv = conv1x1_block(in_channels=in_channels, out_channels=out_channels, activation=lambda_hswish())
        if self.use_se:
            self.se = SEBlock(channels=out_channels, reduction=4, round_mid=True, out_activation=lambda_hsigmoid())

    def _run(self, a):
        y = self.conv(a)
        return self.se(y) if self.use_se else y

    def forward(self, x):
        return engine.boundary(self, x, self._run)


class MobileNetV3Classifier(nn.Module):
    """conv1x1 -> h-swish -> (dropout) -> conv1x1 with bias (reference mobilenetv3.py:134-174); dropout is the identity
    at inference and is not instantiated as a kernel."""
    def __init__(self, in_channels, out_channels, mid_channels, dropout_rate):
        super(MobileNetV3Classifier, self).__init__()
        self.use_dropout = (dropout_rate != 0.0)
        self.conv1 = conv1x1(in_channels=in_channels, out_channels=mid_channels)
        self.activ = HSwish(inplace=True)
        if self.use_dropout:
            self.dropout = nn.Dropout(p=dropout_rate)
        self.conv2 = conv1x1(in_channels=mid_channels, out_channels=out_channels, bias=True)

    def forward(self, x):
        if self.training and self.use_dropout:
            raise RuntimeError("MobileNetV3Classifier: the MI355X path is inference only (call net.eval())")
        y = self.conv1(x, act=engine.act_code(self.activ))
        return self.conv2(y, out_fp32=True)


class MobileNetV3(nn.Module):
    pcv_16bit = "fp16"      # the 16-bit mode "auto" resolves to for this family (engine.compute_dtype_of)

    def __init__(self, channels, exp_channels, init_block_channels, final_block_channels, classifier_mid_channels, kernels3,
                 use_relu, use_se, first_stride, final_use_se, in_channels=3, in_size=(224, 224), num_classes=1000):
        super(MobileNetV3, self).__init__()
        self.in_size = in_size
        self.num_classes = num_classes
        self.features = nn.Sequential()
        self.features.add_module("init_block", conv3x3_block(in_channels=in_channels, out_channels=init_block_channels,
                                                             stride=2, activation=lambda_hswish()))
        in_channels = init_block_channels
        for i, channels_per_stage in enumerate(channels):
            stage = nn.Sequential()
            for j, out_channels in enumerate(channels_per_stage):
                stride = 2 if (j == 0) and ((i != 0) or first_stride) else 1
                stage.add_module("unit{}".format(j + 1), MobileNetV3Unit(
                    in_channels=in_channels, out_channels=out_channels, exp_channels=exp_channels[i][j],
                    use_kernel3=(kernels3[i][j] == 1), stride=stride,
                    activation=(lambda_relu() if use_relu[i][j] == 1 else lambda_hswish()), use_se=(use_se[i][j] == 1)))
                in_channels = out_channels
            self.features.add_module("stage{}".format(i + 1), stage)
        self.features.add_module("final_block", MobileNetV3FinalBlock(in_channels=in_channels, out_channels=final_block_channels,
                                                                      use_se=final_use_se))
        in_channels = final_block_channels
        self.features.add_module("final_pool", AvgPool2dNHWC(kernel_size=7, stride=1, fp32_out=True))
        self.output = MobileNetV3Classifier(in_channels=in_channels, out_channels=num_classes,
                                            mid_channels=classifier_mid_channels, dropout_rate=0.2)
        init_conv_params(self)
        engine.stamp_family_dtype(self)                    # sub-modules called on their own resolve "auto" like the net

    def _head(self, a):
        if a.H != 1 or a.W != 1:
            raise RuntimeError("classifier expects a 1x1 pooled map, got {}x{}".format(a.H, a.W))
        y = self.output(a)
        return y.t.view(y.N, -1)

    def forward(self, x):
        return run_net(self, x, self._head)


# (channels, exp_channels, kernels3, use_relu, use_se, first_stride, final_block_channels) - reference mobilenetv3.py:311-332
_VERSIONS = {
    "small": ([[16], [24, 24], [40, 40, 40, 48, 48], [96, 96, 96]],
              [[16], [72, 88], [96, 240, 240, 120, 144], [288, 576, 576]],
              [[1], [1, 1], [0, 0, 0, 0, 0], [0, 0, 0]],
              [[1], [1, 1], [0, 0, 0, 0, 0], [0, 0, 0]],
              [[1], [0, 0], [1, 1, 1, 1, 1], [1, 1, 1]], True, 576),
    "large": ([[16], [24, 24], [40, 40, 40], [80, 80, 80, 80, 112, 112], [160, 160, 160]],
              [[16], [64, 72], [72, 120, 120], [240, 200, 184, 184, 480, 672], [672, 960, 960]],
              [[1], [1, 1], [0, 0, 0], [1, 1, 1, 1, 1, 1], [0, 0, 0]],
              [[1], [1, 1], [1, 1, 1], [0, 0, 0, 0, 0, 0], [0, 0, 0]],
              [[0], [0, 0], [1, 1, 1], [0, 0, 0, 0, 1, 1], [1, 1, 1]], False, 960),
}


def get_mobilenetv3(version, width_scale, model_name=None, pretrained=False, root=DEFAULT_ROOT, **kwargs):
    if version not in _VERSIONS:
        raise ValueError("Unsupported MobileNetV3 version {}".format(version))
    channels, exp_channels, kernels3, use_relu, use_se, first_stride, final_block_channels = _VERSIONS[version]
    init_block_channels = 16
    if width_scale != 1.0:
        channels = [[round_channels(cij * width_scale) for cij in ci] for ci in channels]
        exp_channels = [[round_channels(cij * width_scale) for cij in ci] for ci in exp_channels]
        init_block_channels = round_channels(init_block_channels * width_scale)
        if width_scale > 1.0:
            final_block_channels = round_channels(final_block_channels * width_scale)
    net = MobileNetV3(channels=channels, exp_channels=exp_channels, init_block_channels=init_block_channels,
                      final_block_channels=final_block_channels, classifier_mid_channels=1280, kernels3=kernels3,
                      use_relu=use_relu, use_se=use_se, first_stride=first_stride, final_use_se=False, **kwargs)
    return maybe_load_pretrained(net, model_name, pretrained, root)


def _variant(version, tag, width_scale):
    name = "mobilenetv3_{}_{}".format(version, tag)

    def factory(**kwargs):
        return get_mobilenetv3(version=version, width_scale=width_scale, model_name=name, **kwargs)
    factory.__name__ = name
    factory.__doc__ = "MobileNetV3 {} x{} (reference mobilenetv3.py:367-595).".format(version, width_scale)
    return factory


for _version in ("small", "large"):
    for _tag, _scale in (("w7d20", 0.35), ("wd2", 0.5), ("w3d4", 0.75), ("w1", 1.0), ("w5d4", 1.25)):
        globals()["mobilenetv3_{}_{}".format(_version, _tag)] = _variant(_version, _tag, _scale)
