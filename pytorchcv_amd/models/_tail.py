"""
    Parameter-free pieces shared by the nets: NHWC pooling modules that stand where the reference has nn.MaxPool2d /
    nn.AvgPool2d (same attribute names, no state), the classifier tail and the pretrained-weights hook of every factory.
"""

__all__ = ['MaxPool2dNHWC', 'AvgPool2dNHWC', 'GlobalAvgPool2dNHWC', 'LinearHead', 'run_net', 'check_channels', 'maybe_load_pretrained', 'init_conv_params']

import os
import torch
import torch.nn as nn
from .. import engine


class MaxPool2dNHWC(nn.Module):
    """nn.MaxPool2d(kernel_size, stride, padding) of ResInitBlock (reference resnet.py:255-258) -> pcv_maxpool2d."""
    def __init__(self, kernel_size, stride, padding, ceil_mode=False):
        super(MaxPool2dNHWC, self).__init__()
        self.kernel_size, self.stride, self.padding, self.ceil_mode = kernel_size, stride, padding, ceil_mode

    def forward(self, x):
        return engine.boundary(self, x, lambda a: engine.maxpool2d(a, self.kernel_size, self.stride, self.padding, self.ceil_mode))


class AvgPool2dNHWC(nn.Module):
    """nn.AvgPool2d(kernel_size, stride) (reference resnet.py:316-318, densenet.py:60) -> pcv_avgpool2d. `fp32_out`: set by the nets
    for their `final_pool` only - the classifier input - which hands fp32 pooled features to the fp32 head (engine.FP32_HEAD);
    anywhere else (DenseNet's transition pools) the map stays in the storage type whatever its size."""
    def __init__(self, kernel_size, stride, fp32_out=False):
        super(AvgPool2dNHWC, self).__init__()
        self.kernel_size, self.stride, self.fp32_out = kernel_size, stride, bool(fp32_out)

    def forward(self, x):
        head = engine.FP32_HEAD and self.fp32_out and isinstance(x, engine.NHWC)
        return engine.boundary(self, x, lambda a: engine.avgpool2d(a, self.kernel_size, self.stride, out_fp32=head))


class GlobalAvgPool2dNHWC(nn.Module):
    """nn.AdaptiveAvgPool2d(output_size=1) (reference efficientnet.py:339) -> pcv_global_avgpool; `fp32_out` as in AvgPool2dNHWC."""
    def __init__(self, output_size=1, fp32_out=False):
        super(GlobalAvgPool2dNHWC, self).__init__()
        if output_size != 1:
            raise NotImplementedError("only AdaptiveAvgPool2d(1) is on the MI355X path")
        self.output_size, self.fp32_out = output_size, bool(fp32_out)

    def forward(self, x):
        head = engine.FP32_HEAD and self.fp32_out and isinstance(x, engine.NHWC)
        return engine.boundary(self, x, lambda a: engine.global_avgpool(a, out_fp32=head))


class LinearHead(nn.Linear):
    """nn.Linear classifier (reference resnet.py:320-322) run as the fused 1x1 GEMM with fp32 logits (pcv_gemm_bias path)."""
    def forward(self, x):
        if getattr(self, "_pcv_runner", None) is None:
            conv = _LinearAsConv(self)
            self._pcv_runner = engine.ConvRunner(conv, None)
        if not isinstance(x, engine.NHWC):
            raise TypeError("LinearHead expects the pooled NHWC handle")
        if x.H != 1 or x.W != 1:
            raise RuntimeError("classifier expects a 1x1 pooled map, got {}x{} (input size must match in_size)".format(x.H, x.W))
        y = self._pcv_runner.run(x, out_fp32=True)
        return y.t.view(y.N, -1)


class _LinearAsConv(object):
    """Duck-typed view of an nn.Linear as a 1x1 nn.Conv2d for ConvRunner (weights are shared, not copied)."""
    def __init__(self, lin):
        self._lin = lin
        self.in_channels, self.out_channels = lin.in_features, lin.out_features
        self.kernel_size, self.stride, self.padding, self.dilation, self.groups = (1, 1), (1, 1), (0, 0), (1, 1), 1
        self.padding_mode = "zeros"

    @property
    def weight(self):
        return self._lin.weight

    @property
    def bias(self):
        return self._lin.bias


def check_channels(net: nn.Module):
    """The MI355X kernels move activations in 16-byte NHWC chunks; channel counts that are not multiples of 8 run with
    zero-padded weights (engine.ConvRunner). The one thing padding cannot express is a grouped (non-depthwise) convolution
    whose channels are not multiples of 8: such a net is refused up front, with the layer named."""
    if getattr(net, "_pcv_channels_ok", False):
        return
    for n, m in net.named_modules():
        if isinstance(m, nn.Conv2d) and 1 < m.groups and not (m.groups == m.in_channels == m.out_channels) and \
                (m.in_channels % 8 or m.out_channels % 8):
            raise NotImplementedError("{}: grouped convolution {} has {} -> {} channels in {} groups; the MI355X path needs "
                                      "multiples of 8 there".format(type(net).__name__, n, m.in_channels, m.out_channels, m.groups))
    net._pcv_channels_ok = True


def run_net(net: nn.Module, x, head, stem: bool = True):
    """Whole-net forward: NCHW fp32 in -> NHWC hot path -> fp32 logits [N, num_classes] out. `stem`: the first convolution
    has stride 2 and takes the image in the 4-channel-padded stem layout (engine.from_nchw)."""
    check_channels(net)
    if isinstance(x, engine.NHWC):
        guard = engine.Fp16Guard(x.device, x.dtype)
        return guard.finish(head(net.features(x)))
    if not torch.is_tensor(x) or x.dim() != 4:
        raise TypeError("expected an NCHW tensor")
    dtype = engine.compute_dtype_of(net)
    guard = engine.Fp16Guard(x.device, engine.DTYPES[dtype][1])      # fp16 only: overflow -> NaN logits, never plausible ones
    a = engine.network_input(x, dtype) if stem else engine.from_nchw(x, dtype, stem=False)
    return guard.finish(head(net.features(a)))


def init_conv_params(net: nn.Module):
    """kaiming_uniform_ on every Conv2d weight, zero conv biases (what each reference net's _init_params does,
    e.g. resnet.py:326-331)."""
    for module in net.modules():
        if isinstance(module, nn.Conv2d):
            nn.init.kaiming_uniform_(module.weight)
            if module.bias is not None:
                nn.init.constant_(module.bias, 0)


def maybe_load_pretrained(net, model_name, pretrained, root):
    if pretrained:
        if (model_name is None) or (not model_name):
            raise ValueError("Parameter `model_name` should be properly initialized for loading pretrained model.")
        from .common.model_store import download_model
        download_model(net=net, model_name=model_name, local_model_store_dir_path=root)
    return net


DEFAULT_ROOT = os.path.join("~", ".torch", "models")
