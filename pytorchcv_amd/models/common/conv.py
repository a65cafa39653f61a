"""
    Convolution blocks with the reference's constructor signatures, attribute names (`conv`, `bn`, `activ` -> identical
    state_dict keys) and factory functions (reference pytorchcv/models/common/conv.py:89-649). `forward` does not run
    Conv2d/BatchNorm2d/activation modules: it issues ONE fused MI355X launch through the C ABI
    (pcv_conv2d_fused / pcv_dwconv2d_fused).
"""

__all__ = ['conv1x1', 'conv3x3', 'depthwise_conv3x3', 'ConvBlock', 'conv1x1_block', 'conv3x3_block', 'conv5x5_block',
           'conv7x7_block', 'dwconv_block', 'dwconv3x3_block', 'dwconv5x5_block', 'DwsConvBlock', 'dwsconv3x3_block', 'BareConv', 'PreConvBlock',
           'pre_conv1x1_block', 'pre_conv3x3_block', 'conv_block_pair', 'conv_block_maxpool', 'mbconv_chain']

import torch.nn as nn
from .activ import lambda_relu, create_activation_layer
from .norm import lambda_batchnorm2d, create_normalization_layer
from ... import engine


class BareConv(nn.Conv2d):
    """nn.Conv2d parameter holder whose forward is the fused MI355X kernel without BN/activation (reference bare
    `conv1x1`, conv.py:89-119: SE convolutions, MobileNetV2 classifier)."""
    def forward(self, x, out_fp32: bool = False, act: int = 0):
        """`act`: engine.act_code of the activation module that follows this convolution in the reference (fused)."""
        if getattr(self, "_pcv_runner", None) is None:
            self._pcv_runner = engine.ConvRunner(self, None)
        return engine.boundary(self, x, lambda a: self._pcv_runner.run(a, act=act, out_fp32=out_fp32))


def conv1x1(in_channels, out_channels, stride=1, groups=1, bias=False):
    return BareConv(in_channels=in_channels, out_channels=out_channels, kernel_size=1, stride=stride, groups=groups, bias=bias)


def conv3x3(in_channels, out_channels, stride=1, padding=1, dilation=1, groups=1, bias=False):
    return BareConv(in_channels=in_channels, out_channels=out_channels, kernel_size=3, stride=stride, padding=padding,
                    dilation=dilation, groups=groups, bias=bias)


def depthwise_conv3x3(channels, stride=1, padding=1, dilation=1, bias=False):
    return BareConv(in_channels=channels, out_channels=channels, kernel_size=3, stride=stride, padding=padding,
                    dilation=dilation, groups=channels, bias=bias)


class ConvBlock(nn.Module):
    """
    Convolution + BatchNorm + activation as one launch (reference ConvBlock, conv.py:204-286).

    Constructor arguments are the reference's. `forward(x, residual=None, post_act=None)` additionally takes the unit's
    skip tensor and the activation that follows the add, so `relu(body(x) + identity)` also stays in the epilogue.
    """
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=False,
                 normalization=lambda_batchnorm2d(), activation=lambda_relu()):
        super(ConvBlock, self).__init__()
        self.normalize = (normalization is not None)
        self.activate = (activation is not None)
        self.use_pad = (isinstance(padding, (list, tuple)) and (len(padding) == 4))
        self._pad4 = tuple(int(p) for p in padding) if self.use_pad else None     # (left, right, top, bottom)
        self.conv = nn.Conv2d(in_channels=in_channels, out_channels=out_channels, kernel_size=kernel_size, stride=stride,
                              padding=(0 if self.use_pad else padding), dilation=dilation, groups=groups, bias=bias)
        if self.normalize:
            self.bn = create_normalization_layer(normalization=normalization, num_features=out_channels)
            if self.bn is None:
                self.normalize = False
            else:
                assert isinstance(self.bn, nn.Module)
        if self.activate:
            self.activ = create_activation_layer(activation)
            if self.activ is None:
                self.activate = False
            else:
                assert isinstance(self.activ, nn.Module)
        self._pcv_runner = None

    def forward(self, x, residual=None, post_act=None, pad4=None, out=None):
        """`out` = (NHWC buffer [N, Ho, Wo, Ctot], channel offset): write the result into that channel slice (a concatenation
        without the copy: common/arch.py `Concurrent`); returns None then."""
        if self._pcv_runner is None:
            self._pcv_runner = engine.ConvRunner(self.conv, self.bn if self.normalize else None, pad4=self._pad4)
        act = engine.act_code(self.activ) if self.activate else 0
        pact = engine.act_code(post_act)
        return engine.boundary(self, x, lambda a: self._pcv_runner.run(a, act=act, residual=residual, post_act=pact,
                                                                       pad4=pad4, out=out))


def conv_block_maxpool(block, x, pool):
    """ConvBlock `block` followed by the MaxPool2dNHWC `pool` (an init block's `conv` -> `pool`): one fused launch when covered
    (pcv_conv2d_maxpool_fused), else the two launches."""
    if isinstance(block, ConvBlock) and isinstance(x, engine.NHWC) and not block.use_pad:
        if block._pcv_runner is None:
            block._pcv_runner = engine.ConvRunner(block.conv, block.bn if block.normalize else None, pad4=block._pad4)
        y = block._pcv_runner.run_maxpool(x, engine.act_code(block.activ) if block.activate else 0, pool.kernel_size,
                                          pool.stride, pool.padding, pool.ceil_mode)
        if y is not None:
            return y
    return pool(block(x))


def conv_block_pair(first, x, residual, post_act, second, id_block=None, x0=None):
    """`first` (with the unit's skip add and activation) and the ConvBlock `second` that consumes its output, as one fused
    launch when the pair of shapes is covered (pcv_conv1x1_pair_fused): (y_first, y_second), else None. With `id_block`
    (the unit's `identity_conv`, a ConvBlock without activation) and the unit input `x0` instead of `residual`, the skip
    tensor is recomputed inside the launch (pcv_conv1x1_pair_idconv_fused)."""
    if not (isinstance(first, ConvBlock) and isinstance(second, ConvBlock) and isinstance(x, engine.NHWC)):
        return None
    for blk in (first, second) + ((id_block,) if id_block is not None else ()):
        if not isinstance(blk, ConvBlock):
            return None
        if blk._pcv_runner is None:
            blk._pcv_runner = engine.ConvRunner(blk.conv, blk.bn if blk.normalize else None, pad4=blk._pad4)
    if id_block is not None:
        if id_block.activate or not isinstance(x0, engine.NHWC):
            return None
        return first._pcv_runner.run_pair_idconv(x, x0, id_block._pcv_runner, engine.act_code(first.activ) if first.activate else 0,
                                                 engine.act_code(post_act), second._pcv_runner,
                                                 engine.act_code(second.activ) if second.activate else 0)
    return first._pcv_runner.run_pair(x, residual, engine.act_code(first.activ) if first.activate else 0,
                                      engine.act_code(post_act), second._pcv_runner,
                                      engine.act_code(second.activ) if second.activate else 0)


def mbconv_chain(exp_block, dw_block, proj_block, x, residual=None, post_act=None):
    """[expand ConvBlock ->] depthwise ConvBlock -> project ConvBlock (+ skip add) as one fused launch when the shapes are
    covered (pcv_mbconv_fused), else None: the caller then runs the blocks one by one."""
    blocks = [b for b in (exp_block, dw_block, proj_block) if b is not None]
    if not isinstance(x, engine.NHWC) or not all(isinstance(b, ConvBlock) for b in blocks):
        return None
    for blk in blocks:
        if blk._pcv_runner is None:
            blk._pcv_runner = engine.ConvRunner(blk.conv, blk.bn if blk.normalize else None, pad4=blk._pad4)

    def code(blk):
        return engine.act_code(blk.activ) if blk.activate else 0
    return engine.mbconv_fused(exp_block._pcv_runner if exp_block is not None else None,
                               code(exp_block) if exp_block is not None else 0, dw_block._pcv_runner, code(dw_block),
                               proj_block._pcv_runner, code(proj_block), x, residual, engine.act_code(post_act))


def conv1x1_block(padding=0, **kwargs):
    return ConvBlock(kernel_size=1, padding=padding, **kwargs)


def conv3x3_block(padding=1, **kwargs):
    return ConvBlock(kernel_size=3, padding=padding, **kwargs)


def conv5x5_block(padding=2, **kwargs):
    return ConvBlock(kernel_size=5, padding=padding, **kwargs)


def conv7x7_block(padding=3, **kwargs):
    return ConvBlock(kernel_size=7, padding=padding, **kwargs)


def dwconv_block(out_channels, padding=1, **kwargs):
    return ConvBlock(out_channels=out_channels, padding=padding, groups=out_channels, **kwargs)


def dwconv3x3_block(padding=1, **kwargs):
    return dwconv_block(kernel_size=3, padding=padding, **kwargs)


def dwconv5x5_block(padding=2, **kwargs):
    return dwconv_block(kernel_size=5, padding=padding, **kwargs)


class DwsConvBlock(nn.Module):
    """Depthwise-separable block = depthwise ConvBlock + pointwise ConvBlock (reference conv.py:546-618)."""
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, bias=False,
                 dw_normalization=lambda_batchnorm2d(), pw_normalization=lambda_batchnorm2d(), dw_activation=lambda_relu(),
                 pw_activation=lambda_relu()):
        super(DwsConvBlock, self).__init__()
        self.dw_conv = dwconv_block(in_channels=in_channels, out_channels=in_channels, kernel_size=kernel_size, stride=stride,
                                    padding=padding, dilation=dilation, bias=bias, normalization=dw_normalization,
                                    activation=dw_activation)
        self.pw_conv = conv1x1_block(in_channels=in_channels, out_channels=out_channels, bias=bias,
                                     normalization=pw_normalization, activation=pw_activation)

    def _run(self, a):
        y = mbconv_chain(None, self.dw_conv, self.pw_conv, a)
        return y if y is not None else self.pw_conv(self.dw_conv(a))

    def forward(self, x):
        return engine.boundary(self, x, self._run)


def dwsconv3x3_block(padding=1, **kwargs):
    return DwsConvBlock(kernel_size=3, padding=padding, **kwargs)


class PreConvBlock(nn.Module):
    """
    BatchNorm + activation + convolution, the pre-activation order (reference PreConvBlock, conv.py:652-786): same
    constructor, same attributes (`bn`, `activ`, `conv`), `forward` returns `x` or `(x, x_pre_activ)`.

    On the MI355X path BN+activation is an epilogue wherever a convolution produces the tensor it applies to, so a chain
    `PreConvBlock -> PreConvBlock` runs as `preact(x)` once (pcv_bn_act), then `conv_then(a, next_block)` per block: this
    block's convolution with the NEXT block's BN+activation in its epilogue. `forward` is the unfused block for drop-in use.
    """
    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, dilation=1, bias=False,
                 normalization=lambda_batchnorm2d(), activation=lambda_relu(), return_preact=False):
        super(PreConvBlock, self).__init__()
        self.normalize = (normalization is not None)
        self.activate = (activation is not None)
        self.return_preact = return_preact
        if self.normalize:
            self.bn = create_normalization_layer(normalization=normalization, num_features=in_channels)
            if self.bn is None:
                self.normalize = False
            else:
                assert isinstance(self.bn, nn.BatchNorm2d)
        if self.activate:
            self.activ = create_activation_layer(activation)
            if self.activ is None:
                self.activate = False
            else:
                assert isinstance(self.activ, nn.Module)
        self.conv = nn.Conv2d(in_channels=in_channels, out_channels=out_channels, kernel_size=kernel_size, stride=stride,
                              padding=padding, dilation=dilation, bias=bias)
        self._pcv_pre = None
        self._pcv_runners = {}

    def act_code(self):
        return engine.act_code(self.activ) if self.activate else 0

    def preact(self, a):
        """activ(bn(a)) as its own launch."""
        if not self.normalize:
            raise NotImplementedError("PreConvBlock without normalization on the MI355X path")
        if self._pcv_pre is None:
            self._pcv_pre = engine.BnActRunner(self.bn)
        return self._pcv_pre.run(a, self.act_code())

    def conv_then(self, a, next_block=None, residual=None, out=None, se=None):
        """This block's convolution applied to the already pre-activated `a`; `next_block`'s BN + activation (the
        pre-activation of the following PreConvBlock) ride in the epilogue, or `residual` is added (last block of a unit)."""
        key = id(next_block)
        if key not in self._pcv_runners:
            bn = next_block.bn if (next_block is not None and next_block.normalize) else None
            self._pcv_runners[key] = engine.ConvRunner(self.conv, bn)
        act = next_block.act_code() if next_block is not None else 0
        runner = self._pcv_runners[key]
        gate = None
        c = self.conv
        if se is not None and next_block is None and out is None and engine.FUSE_UNITS and tuple(c.kernel_size) == (1, 1) and \
                tuple(c.stride) == (1, 1) and tuple(c.padding) == (0, 0) and a.dense:
            # SEBlock `se` follows this (linear) convolution: its gate is computed from the squeezed INPUT and applied in the epilogue
            w1, b1, w2, b2 = se._mlp()
            gate = runner.squeezed_excite(a, w1, b1, w2, b2, engine.act_code(se.activ), engine.act_code(se.sigmoid))
        elif se is not None:
            return se(runner.run(a, act=act, out=out), residual=residual)
        return runner.run(a, act=act, residual=residual, out=out, gate=gate)

    def _run(self, a):
        pre = self.preact(a) if (self.normalize or self.activate) else a
        y = self.conv_then(pre)
        return (y, pre) if self.return_preact else y

    def forward(self, x):
        return engine.boundary(self, x, self._run)      # (tensor entry: the module's resolved type + the fp16 range guard)


def pre_conv1x1_block(stride=1, padding=0, **kwargs):
    return PreConvBlock(kernel_size=1, stride=stride, padding=padding, **kwargs)


def pre_conv3x3_block(stride=1, padding=1, **kwargs):
    return PreConvBlock(kernel_size=3, stride=stride, padding=padding, **kwargs)

