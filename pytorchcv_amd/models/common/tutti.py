"""
    Small building blocks of the reference's `common/tutti.py` on the NHWC hot path: stand-alone BatchNorm + activation
    (`NormActivation`, tutti.py:157-191), bilinear / nearest resampling (`InterpolationBlock`, tutti.py:194-264) and the channel
    shuffle (`ChannelShuffle`, tutti.py:267-321).
"""

__all__ = ['NormActivation', 'InterpolationBlock', 'ChannelShuffle', 'channel_shuffle']

import torch.nn as nn
from ... import engine
from .activ import lambda_relu, create_activation_layer
from .norm import lambda_batchnorm2d, create_normalization_layer


class NormActivation(nn.Module):
    """BatchNorm2d followed by an activation, on its own (the closing block of pre-activation trunks): one elementwise launch
    with the folded scale / shift (pcv_bn_act). Registers `bn` and `activ` like the reference's block."""
    def __init__(self, in_channels, normalization=lambda_batchnorm2d(), activation=lambda_relu()):
        super(NormActivation, self).__init__()
        self.bn = create_normalization_layer(normalization=normalization, num_features=in_channels)
        assert isinstance(self.bn, nn.BatchNorm2d)
        self.activ = create_activation_layer(activation)
        assert isinstance(self.activ, nn.Module)
        self._pcv_pre = None

    def _run(self, a):
        if self._pcv_pre is None:
            self._pcv_pre = engine.BnActRunner(self.bn)
        return self._pcv_pre.run(a, engine.act_code(self.activ))

    def forward(self, x):
        return engine.boundary(self, x, self._run)


class InterpolationBlock(nn.Module):
    """F.interpolate as a module: output size = `out_size`, or the input size times (`up`) / divided by `scale_factor`, or the
    `size` given at call time; `mode` "bilinear" (with `align_corners`) or "nearest" (pcv_interpolate)."""
    def __init__(self, scale_factor, out_size=None, mode="bilinear", align_corners=True, up=True):
        super(InterpolationBlock, self).__init__()
        self.scale_factor = scale_factor
        self.out_size = out_size
        self.mode = mode
        self.align_corners = align_corners
        self.up = up

    def calc_out_size(self, x):
        if self.out_size is not None:
            return tuple(self.out_size)
        h, w = (x.H, x.W) if isinstance(x, engine.NHWC) else tuple(x.shape[2:])
        if self.up:
            return (h * self.scale_factor, w * self.scale_factor)
        return (h // self.scale_factor, w // self.scale_factor)

    def forward(self, x, size=None):
        """Reference tutti.py:225-238: bilinear (or an explicit `size`) interpolates to calc_out_size(x) / `size`; any other mode
        without `size` calls F.interpolate(scale_factor=...), which ignores `out_size` and `up` - the output is the input size times
        `scale_factor` - and, like torch, refuses an `align_corners` that is not None."""
        if self.mode not in ("bilinear", "nearest"):
            raise NotImplementedError("InterpolationBlock mode {} is not on the MI355X path".format(self.mode))
        bilinear = self.mode == "bilinear"
        if not bilinear and self.align_corners is not None:
            raise ValueError("align_corners option can only be set with the interpolating modes: linear | bilinear | bicubic | trilinear")
        if bilinear or size is not None:
            out_size = tuple(size) if size is not None else self.calc_out_size(x)
        else:
            h, w = (x.H, x.W) if isinstance(x, engine.NHWC) else tuple(x.shape[2:])
            oh, ow = h * self.scale_factor, w * self.scale_factor
            if oh != int(oh) or ow != int(ow) or oh < 1 or ow < 1:
                # (F.interpolate would floor the size but keep 1 / scale_factor as the source step: not the size-derived step of
                # pcv_interpolate)
                raise NotImplementedError("nearest interpolation by scale_factor {} of a {}x{} map: only whole output sizes are on the "
                                          "MI355X path".format(self.scale_factor, h, w))
            out_size = (int(oh), int(ow))
        return engine.boundary(self, x, lambda a: engine.interpolate(a, out_size, bilinear, bool(self.align_corners) if bilinear else False))

    def __repr__(self):
        return "{}(scale_factor={}, out_size={}, mode={}, align_corners={}, up={})".format(
            type(self).__name__, self.scale_factor, self.out_size, self.mode, self.align_corners, self.up)


def channel_shuffle(x, groups, module=None):
    """Channel shuffle of ShuffleNet: channels viewed as [groups, C / groups] and transposed. On the hot path for groups == 2
    (the interleave of the two halves, pcv_channel_interleave2). `module`: whose compute dtype an NCHW input is converted to."""
    def run(a):
        if groups != 2 or a.C % 2:
            raise NotImplementedError("channel_shuffle with {} groups is not on the MI355X path (groups = 2 is)".format(groups))
        half = a.C // 2
        second = engine.NHWC(a.t[:, :, :, half:], a.N, a.H, a.W, half, cpitch=a.cpitch)     # a view: same pitch, offset pointer
        return engine.cat_shuffle2(a, second, half)
    if isinstance(x, engine.NHWC):
        return run(x)
    return engine.boundary(module if module is not None else nn.Module(), x, run)


class ChannelShuffle(nn.Module):
    def __init__(self, channels, groups):
        super(ChannelShuffle, self).__init__()
        if channels % groups != 0:
            raise ValueError("channels must be divisible by groups")
        self.groups = groups

    def forward(self, x):
        return channel_shuffle(x, self.groups, module=self)

    def __repr__(self):
        return "{}(groups={})".format(type(self).__name__, self.groups)
