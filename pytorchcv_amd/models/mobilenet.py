"""
    MobileNet (v1) for ImageNet-1K on the MI355X hot path (reference pytorchcv/models/mobilenet.py:17-250): a 3x3/2 stem and
    13 depthwise-separable units, each = depthwise 3x3 kernel + pointwise 1x1 MFMA kernel, both with fused BN + ReLU.
"""

__all__ = ['MobileNet', 'get_mobilenet', 'mobilenet_w1', 'mobilenet_w3d4', 'mobilenet_wd2', 'mobilenet_wd4']

import torch.nn as nn
from .common.activ import lambda_relu
from .common.norm import lambda_batchnorm2d
from .common.conv import conv3x3_block, dwsconv3x3_block
from ._tail import AvgPool2dNHWC, LinearHead, run_net, maybe_load_pretrained, DEFAULT_ROOT


class MobileNet(nn.Module):
    def __init__(self, channels, first_stage_stride, dw_use_bn=True, dw_activation=lambda_relu(), in_channels=3,
                 in_size=(224, 224), num_classes=1000):
        super(MobileNet, self).__init__()
        self.in_size = in_size
        self.num_classes = num_classes
        dw_normalization = lambda_batchnorm2d() if dw_use_bn else None
        self.features = nn.Sequential()
        init_block_channels = channels[0][0]
        self.features.add_module("init_block", conv3x3_block(in_channels=in_channels, out_channels=init_block_channels, stride=2))
        in_channels = init_block_channels
        for i, channels_per_stage in enumerate(channels[1:]):
            stage = nn.Sequential()
            for j, out_channels in enumerate(channels_per_stage):
                stride = 2 if (j == 0) and ((i != 0) or first_stage_stride) else 1
                stage.add_module("unit{}".format(j + 1), dwsconv3x3_block(
                    in_channels=in_channels, out_channels=out_channels, stride=stride, dw_normalization=dw_normalization,
                    dw_activation=dw_activation))
                in_channels = out_channels
            self.features.add_module("stage{}".format(i + 1), stage)
        self.features.add_module("final_pool", AvgPool2dNHWC(kernel_size=7, stride=1))
        self.output = LinearHead(in_features=in_channels, out_features=num_classes)
        self._init_params()

    def _init_params(self):
        # same initialisation families as the reference (mobilenet.py:78-90)
        for name, module in self.named_modules():
            if "dw_conv.conv" in name:
                nn.init.kaiming_normal_(module.weight, mode="fan_in")
            elif name == "init_block.conv" or "pw_conv.conv" in name:
                nn.init.kaiming_normal_(module.weight, mode="fan_out")
            elif "bn" in name:
                nn.init.constant_(module.weight, 1)
                nn.init.constant_(module.bias, 0)
            elif "output" in name:
                nn.init.kaiming_normal_(module.weight, mode="fan_out")
                nn.init.constant_(module.bias, 0)

    def forward(self, x):
        return run_net(self, x, self.output)


def get_mobilenet(width_scale, dws_simplified=False, model_name=None, pretrained=False, root=DEFAULT_ROOT, **kwargs):
    channels = [[32], [64], [128, 128], [256, 256], [512] * 6, [1024, 1024]]
    if width_scale != 1.0:
        channels = [[int(c * width_scale) for c in ci] for ci in channels]
    net = MobileNet(channels=channels, first_stage_stride=False, dw_use_bn=not dws_simplified,
                    dw_activation=None if dws_simplified else lambda_relu(), **kwargs)
    return maybe_load_pretrained(net, model_name, pretrained, root)


def mobilenet_w1(**kwargs):
    return get_mobilenet(width_scale=1.0, model_name="mobilenet_w1", **kwargs)


def mobilenet_w3d4(**kwargs):
    return get_mobilenet(width_scale=0.75, model_name="mobilenet_w3d4", **kwargs)


def mobilenet_wd2(**kwargs):
    return get_mobilenet(width_scale=0.5, model_name="mobilenet_wd2", **kwargs)


def mobilenet_wd4(**kwargs):
    return get_mobilenet(width_scale=0.25, model_name="mobilenet_wd4", **kwargs)
