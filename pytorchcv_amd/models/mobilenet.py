"""
    MobileNet (v1) for ImageNet-1K on the MI355X hot path (architecture: reference pytorchcv/models/mobilenet.py:17-250): a 3x3/2
    stem and 13 depthwise-separable units, each = depthwise 3x3 kernel + pointwise 1x1 MFMA kernel, both with fused BN + ReLU.
"""

__all__ = ['MobileNet', 'get_mobilenet']

import torch.nn as nn
from .common.activ import lambda_relu
from .common.norm import lambda_batchnorm2d
from .common.conv import conv3x3_block, dwsconv3x3_block
from ._build import ClassifierNet, add_stages, register_variants, scale_widths
from ._tail import maybe_load_pretrained, DEFAULT_ROOT

# stem width, then the widths of the depthwise-separable units stage by stage (a new stage = stride 2)
_WIDTHS = [[32], [64], [128, 128], [256, 256], [512] * 6, [1024, 1024]]
# weight-initialisation family per parameter-name fragment (what the reference's _init_params does, mobilenet.py:78-90)
_INIT = (("dw_conv.conv", "fan_in"), ("pw_conv.conv", "fan_out"), ("init_block.conv", "fan_out"), ("output", "fan_out"))


class MobileNet(ClassifierNet):
    def __init__(self, channels, first_stage_stride, dw_use_bn=True, dw_activation=lambda_relu(), in_channels=3,
                 in_size=(224, 224), num_classes=1000):
        super(MobileNet, self).__init__(in_size, num_classes)
        dw_norm = lambda_batchnorm2d() if dw_use_bn else None
        stem = channels[0][0]
        self.features.add_module("init_block", conv3x3_block(in_channels=in_channels, out_channels=stem, stride=2))
        width = add_stages(
            self.features, stem, channels[1:], downsample_first=first_stage_stride,
            make_unit=lambda cin, cout, stride, i, j: dwsconv3x3_block(
                in_channels=cin, out_channels=cout, stride=stride, dw_normalization=dw_norm, dw_activation=dw_activation))
        self.finish(width, init=MobileNet._init_params)

    def _init_params(self):
        for name, module in self.named_modules():
            mode = next((m for frag, m in _INIT if frag in name and hasattr(module, "weight")), None)
            if isinstance(module, nn.BatchNorm2d):
                nn.init.constant_(module.weight, 1)
                nn.init.constant_(module.bias, 0)
            elif mode is not None and isinstance(module, (nn.Conv2d, nn.Linear)):
                nn.init.kaiming_normal_(module.weight, mode=mode)
                if module.bias is not None:
                    nn.init.constant_(module.bias, 0)


def get_mobilenet(width_scale, dws_simplified=False, model_name=None, pretrained=False, root=DEFAULT_ROOT, **kwargs):
    """`dws_simplified` (FD-MobileNet style): no BN / activation between the depthwise and the pointwise convolution."""
    net = MobileNet(channels=scale_widths(_WIDTHS, width_scale), first_stage_stride=False, dw_use_bn=not dws_simplified,
                    dw_activation=None if dws_simplified else lambda_relu(), **kwargs)
    return maybe_load_pretrained(net, model_name, pretrained, root)


register_variants(__name__, get_mobilenet, {
    "mobilenet_w1": dict(width_scale=1.0), "mobilenet_w3d4": dict(width_scale=0.75),
    "mobilenet_wd2": dict(width_scale=0.5), "mobilenet_wd4": dict(width_scale=0.25)})
