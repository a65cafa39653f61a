"""
    Squeeze-and-Excitation (reference pytorchcv/models/common/att.py:15-105): same attributes (`conv1`/`conv2` with bias, or
    `fc1`/`fc2`), three small launches on the hot path: spatial mean -> fp32 excitation MLP -> channel scale fused with the
    unit's residual add and activation.
"""

__all__ = ['round_channels', 'SEBlock']

import torch.nn as nn
from .activ import lambda_relu, lambda_sigmoid, create_activation_layer
from .conv import conv1x1
from ... import engine


def round_channels(channels, divisor=8):
    """Make-divisible rule of reference att.py:15-35."""
    rounded = max(int(channels + divisor / 2.0) // divisor * divisor, divisor)
    if float(rounded) < 0.9 * channels:
        rounded += divisor
    return rounded


class SEBlock(nn.Module):
    def __init__(self, channels, reduction=16, mid_channels=None, round_mid=False, use_conv=True,
                 mid_activation=lambda_relu(), out_activation=lambda_sigmoid()):
        super(SEBlock, self).__init__()
        self.use_conv = use_conv
        if mid_channels is None:
            mid_channels = channels // reduction if not round_mid else round_channels(float(channels) / reduction)
        self.pool = nn.AdaptiveAvgPool2d(output_size=1)      # marker only; the squeeze kernel computes it
        if use_conv:
            self.conv1 = conv1x1(in_channels=channels, out_channels=mid_channels, bias=True)
        else:
            self.fc1 = nn.Linear(in_features=channels, out_features=mid_channels)
        self.activ = create_activation_layer(mid_activation)
        if use_conv:
            self.conv2 = conv1x1(in_channels=mid_channels, out_channels=channels, bias=True)
        else:
            self.fc2 = nn.Linear(in_features=mid_channels, out_features=channels)
        self.sigmoid = create_activation_layer(out_activation)

    def _mlp(self):
        a, b = (self.conv1, self.conv2) if self.use_conv else (self.fc1, self.fc2)
        w1 = a.weight.detach().float().reshape(a.weight.shape[0], -1).contiguous()
        w2 = b.weight.detach().float().reshape(b.weight.shape[0], -1).contiguous()
        return w1, a.bias.detach().float().contiguous(), w2, b.bias.detach().float().contiguous()

    def run_behind(self, conv_block, z, residual=None, post_act=None, next_conv=None):
        """`self(conv_block(z), residual, post_act)` in ONE pass over the wide tensor when `conv_block` is a plain 1x1 ConvBlock
        without activation (the bottleneck's last convolution, seresnet.py:60-71): BN(conv(.)) is affine, so the squeeze
        mean_hw(BN(conv(z))) = BN(conv(mean_hw(z))) is taken on the narrower input z, the excitation runs BEFORE the convolution
        and the channel scale + skip add + activation ride in its epilogue (pcv_conv2d_gated_fused). Returns None when the block
        is not of that shape (the caller then runs conv_block and this module one after the other). With `next_conv` (the next
        unit's first 1x1 ConvBlock) the result may be the tuple (y, next_conv(y)) of pcv_conv1x1_pair_gated_fused."""
        from .conv import ConvBlock
        c = getattr(conv_block, "conv", None)
        if not (engine.FUSE_UNITS and isinstance(conv_block, ConvBlock) and isinstance(z, engine.NHWC) and not conv_block.activate and
                not conv_block.use_pad and c is not None and tuple(c.kernel_size) == (1, 1) and tuple(c.stride) == (1, 1) and
                tuple(c.padding) == (0, 0) and c.groups == 1 and z.dense):
            return None
        if conv_block._pcv_runner is None:
            conv_block._pcv_runner = engine.ConvRunner(conv_block.conv, conv_block.bn if conv_block.normalize else None)
        runner = conv_block._pcv_runner
        w1, b1, w2, b2 = self._mlp()
        gate = runner.squeezed_excite(z, w1, b1, w2, b2, engine.act_code(self.activ), engine.act_code(self.sigmoid))
        if next_conv is not None and residual is not None and isinstance(next_conv, ConvBlock):
            # ... and the next unit's first 1x1 in the same launch: returns (y, next conv1 output)
            if next_conv._pcv_runner is None:
                next_conv._pcv_runner = engine.ConvRunner(next_conv.conv, next_conv.bn if next_conv.normalize else None, pad4=next_conv._pad4)
            pair = runner.run_pair(z, residual, 0, engine.act_code(post_act), next_conv._pcv_runner,
                                   engine.act_code(next_conv.activ) if next_conv.activate else 0, gate=gate)
            if pair is not None:
                return pair
        return runner.run(z, act=0, residual=residual, post_act=engine.act_code(post_act), gate=gate)

    def forward(self, x, residual=None, post_act=None):
        w1, b1, w2, b2 = self._mlp()
        return engine.boundary(self, x, lambda a: engine.se_forward(
            a, w1, b1, w2, b2, engine.act_code(self.activ), engine.act_code(self.sigmoid), residual, engine.act_code(post_act)))
