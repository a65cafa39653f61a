"""
    MobileNetV2 for ImageNet-1K on the MI355X hot path (reference pytorchcv/models/mobilenetv2.py:16-220): expand 1x1 (MFMA),
    depthwise 3x3 (direct HBM-bound kernel), project 1x1 with the skip add in its epilogue; bias-free 1x1-conv classifier.
"""

__all__ = ['MobileNetV2', 'LinearBottleneck', 'get_mobilenetv2']

import torch.nn as nn
from .common.activ import lambda_relu6
from .common.conv import conv1x1, conv1x1_block, conv3x3_block, dwconv3x3_block, mbconv_chain
from ._build import ClassifierNet, add_stages, register_variants, scale_widths
from ._tail import maybe_load_pretrained, DEFAULT_ROOT
from .. import engine


class LinearBottleneck(nn.Module):
    """Inverted residual unit (reference mobilenetv2.py:16-71); no activation after the add."""
    def __init__(self, in_channels, out_channels, stride, expansion, remove_exp_conv, activation):
        super(LinearBottleneck, self).__init__()
        self.residual = (in_channels == out_channels) and (stride == 1)
        mid_channels = in_channels * 6 if expansion else in_channels
        self.use_exp_conv = (expansion or (not remove_exp_conv))
        if self.use_exp_conv:
            self.conv1 = conv1x1_block(in_channels=in_channels, out_channels=mid_channels, activation=activation)
        self.conv2 = dwconv3x3_block(in_channels=mid_channels, out_channels=mid_channels, stride=stride, activation=activation)
        self.conv3 = conv1x1_block(in_channels=mid_channels, out_channels=out_channels, activation=None)

    def _run(self, a):
        residual = a if self.residual else None
        y = mbconv_chain(self.conv1 if self.use_exp_conv else None, self.conv2, self.conv3, a, residual=residual)
        if y is not None:
            return y                                     # the whole unit was one launch (csrc/mbconv.hpp)
        y = self.conv1(a) if self.use_exp_conv else a
        return self.conv3(self.conv2(y), residual=residual)

    def forward(self, x):
        return engine.boundary(self, x, self._run)


class MobileNetV2(ClassifierNet):
    pcv_16bit = "fp16"      # the 16-bit mode "auto" resolves to for this family (engine.compute_dtype_of)

    def __init__(self, channels, init_block_channels, final_block_channels, remove_exp_conv, in_channels=3,
                 in_size=(224, 224), num_classes=1000):
        super(MobileNetV2, self).__init__(in_size, num_classes)
        act = lambda_relu6()
        self.features.add_module("init_block", conv3x3_block(in_channels=in_channels, out_channels=init_block_channels,
                                                             stride=2, activation=act))
        width = add_stages(
            self.features, init_block_channels, channels,
            make_unit=lambda cin, cout, stride, i, j: LinearBottleneck(
                in_channels=cin, out_channels=cout, stride=stride, expansion=(i, j) != (0, 0),   # the very first unit does not expand
                remove_exp_conv=remove_exp_conv, activation=act))
        self.features.add_module("final_block", conv1x1_block(in_channels=width, out_channels=final_block_channels, activation=act))
        self.finish(final_block_channels, head=conv1x1(in_channels=final_block_channels, out_channels=num_classes, bias=False))
        engine.stamp_family_dtype(self)                    # sub-modules called on their own resolve "auto" like the net

    def _head(self, a):
        if a.H != 1 or a.W != 1:
            raise RuntimeError("classifier expects a 1x1 pooled map, got {}x{}".format(a.H, a.W))
        y = self.output(a, out_fp32=True)
        return y.t.view(y.N, -1)


# (width, units, opens a new stage = stride 2): seven groups of inverted-residual units in five stages
_GROUPS = ((16, 1, False), (24, 2, True), (32, 3, True), (64, 4, True), (96, 3, False), (160, 3, True), (320, 1, False))
_STEM, _FINAL = 32, 1280


def get_mobilenetv2(width_scale, remove_exp_conv=False, model_name=None, pretrained=False, root=DEFAULT_ROOT, **kwargs):
    stages = []
    for width, units, new_stage in _GROUPS:
        if new_stage or not stages:
            stages.append([])
        stages[-1] += [width] * units
    net = MobileNetV2(channels=scale_widths(stages, width_scale),
                      init_block_channels=_STEM if width_scale == 1.0 else int(_STEM * width_scale),
                      final_block_channels=int(_FINAL * width_scale) if width_scale > 1.0 else _FINAL,   # never narrower than 1280
                      remove_exp_conv=remove_exp_conv, **kwargs)
    return maybe_load_pretrained(net, model_name, pretrained, root)


register_variants(__name__, get_mobilenetv2, {
    "mobilenetv2_w1": dict(width_scale=1.0), "mobilenetv2_w3d4": dict(width_scale=0.75),
    "mobilenetv2_wd2": dict(width_scale=0.5), "mobilenetv2_wd4": dict(width_scale=0.25)})
