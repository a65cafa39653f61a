"""
    ORACLE - test infrastructure, not product code.

    Block-level CPU restatement: forward of one reference block (by constructor name + kwargs, as listed in
    tests/golden/cases.py) over a state_dict, in reference-exact fp32 mode or quantisation-matched 16-bit mode
    (see oracle/refnet.py). Reference lines: common/conv.py:204-543 (ConvBlock and factories), common/att.py:94-105
    (SEBlock), resnet.py:143-263 (ResUnit, ResInitBlock), mobilenetv2.py:16-71 (LinearBottleneck), resnext.py:17-116
    (ResNeXtUnit), seresnet.py:17-72 (SEResUnit), mobilenetv3.py:18-93 (MobileNetV3Unit),
    efficientnet.py:58-239 (EffiDwsConvUnit, EffiInvResUnit, EffiInitBlock), conv.py:652-810 (PreConvBlock),
    preresnet.py:109-222 (PreResUnit, PreResInitBlock, PreResActivation), densenet.py:16-91 (DenseUnit, TransitionBlock),
    shufflenetv2.py:17-120 (ShuffleUnit, ShuffleInitBlock).
"""

__all__ = ['block_forward', 'f4_block_forward']

import torch
import torch.nn.functional as F
from .refnet import (Quant, conv_block, se_block, conv_then_se, _res_body, tf_same_pad, effi_dws_unit, effi_inv_res_unit, bn_act,
                     pre_conv_chain, preres_unit, preres_init_block, dense_unit, dense_transition, shuffle_unit)

_KSIZE = {"conv1x1_block": (1, 0), "conv3x3_block": (3, 1), "conv5x5_block": (5, 2), "conv7x7_block": (7, 3),
          "dwconv3x3_block": (3, 1), "dwconv5x5_block": (5, 2)}


def _act_name(a, default="relu"):
    if a is None:
        return None
    if isinstance(a, str):
        return a
    return default


def _conv_kind(kind, kw, sd, x, q, prefix=""):
    kw = dict(kw)
    if kind == "ConvBlock":
        k = kw["kernel_size"]
        pad = kw.get("padding", 0)
    else:
        k, pad = _KSIZE[kind]
        pad = kw.get("padding", pad)
    groups = kw.get("groups", 1)
    if kind.startswith("dwconv"):
        groups = kw["out_channels"]
    act = _act_name(kw["activation"]) if "activation" in kw else "relu"
    normalize = not ("normalization" in kw and kw["normalization"] is None)
    if isinstance(pad, (list, tuple)) and len(pad) == 4:
        pad = tuple(pad)
    return conv_block(sd, prefix, x, stride=kw.get("stride", 1), padding=pad, dilation=kw.get("dilation", 1),
                      groups=groups, act=act, q=q, normalize=normalize)


def block_forward(kind: str, kwargs: dict, sd: dict, x: torch.Tensor, quant: str | None = None) -> torch.Tensor:
    q = Quant(quant)
    kw = dict(kwargs)
    with torch.no_grad():
        x = q.r(x.float())
        if kind in _KSIZE or kind == "ConvBlock":
            return _conv_kind(kind, kw, sd, x, q)
        if kind == "SEBlock":
            return se_block(sd, "", x, q=q, mid_act=_act_name(kw.get("mid_activation", "relu")),
                            out_act=_act_name(kw.get("out_activation", "sigmoid"), "sigmoid"))
        if kind == "MobileNetV3Unit":
            # mobilenetv3.py:40-93
            stride, act = kw["stride"], _act_name(kw["activation"])
            residual = x if (kw["in_channels"] == kw["out_channels"] and stride == 1) else None
            y = x
            if kw["exp_channels"] != kw["out_channels"]:
                y = conv_block(sd, "exp_conv.", y, act=act, q=q)
            k = 3 if kw["use_kernel3"] else 5
            y = conv_block(sd, "conv1.", y, stride=stride, padding=k // 2, groups=y.shape[1], act=act, q=q)
            if kw["use_se"]:
                y = se_block(sd, "se.", y, q=q, out_act="hsigmoid")
            return conv_block(sd, "conv2.", y, act=None, q=q, residual=residual)
        if kind == "EffiInitBlock":
            # efficientnet.py:235-239
            pad = tf_same_pad(x.shape[2], x.shape[3], 3, 2) if kw["tf_mode"] else 1
            return conv_block(sd, "conv.", x, stride=2, padding=pad, act="swish", q=q, eps=kw["bn_eps"])
        if kind == "EffiDwsConvUnit":
            return effi_dws_unit(sd, "", x, q, kw["tf_mode"], kw["bn_eps"], residual_ok=(kw["stride"] == 1))
        if kind == "EffiInvResUnit":
            return effi_inv_res_unit(sd, "", x, q, kw["stride"], kw["tf_mode"], kw["bn_eps"])
        if kind in ("pre_conv3x3_block", "pre_conv1x1_block"):
            # PreConvBlock.forward, conv.py:776-786 (return_preact=False)
            return pre_conv_chain(sd, [""], [kw.get("stride", 1)], bn_act(sd, "bn.", x, q), q)
        if kind == "PreResUnit":
            return preres_unit(sd, "", x, kw["stride"], kw["bottleneck"], kw["conv1_stride"], q)
        if kind == "PreResInitBlock":
            return preres_init_block(sd, "", x, q)
        if kind == "PreResActivation":
            return bn_act(sd, "bn.", x, q)
        if kind == "ShuffleUnit":
            return shuffle_unit(sd, "", x, kw["downsample"], q)
        if kind == "ShuffleInitBlock":
            y = conv_block(sd, "conv.", x, stride=2, padding=1, q=q)
            return F.max_pool2d(y, kernel_size=3, stride=2, padding=0, ceil_mode=True)
        if kind == "DenseUnit":
            return dense_unit(sd, "", x, q)
        if kind == "TransitionBlock":
            return dense_transition(sd, "", x, q)
        if kind == "ResInitBlock":
            y = conv_block(sd, "conv.", x, stride=2, padding=3, q=q)
            return F.max_pool2d(y, kernel_size=3, stride=2, padding=1)
        if kind in ("ResUnit", "SEResUnit"):
            stride = kw.get("stride", 1)
            resize = (kw["in_channels"] != kw["out_channels"]) or (stride != 1)
            identity = conv_block(sd, "identity_conv.", x, stride=stride, act=None, q=q) if resize else x
            if kind == "SEResUnit" and kw["bottleneck"]:
                y = conv_block(sd, "body.conv1.", x, stride=(stride if kw["conv1_stride"] else 1), q=q)
                y = conv_block(sd, "body.conv2.", y, stride=(1 if kw["conv1_stride"] else stride), padding=1, q=q)
                return conv_then_se(sd, "body.conv3.", "se.", y, q, identity, "relu")
            if kind == "SEResUnit":
                y = _res_body(sd, "body.", x, stride, kw["bottleneck"], kw["conv1_stride"], q, None, None)
                return se_block(sd, "se.", y, q=q, residual=identity, post_act="relu")
            return _res_body(sd, "body.", x, stride, kw.get("bottleneck", True), kw.get("conv1_stride", False), q,
                             identity, "relu")
        if kind == "LinearBottleneck":
            stride = kw["stride"]
            residual = x if (kw["in_channels"] == kw["out_channels"] and stride == 1) else None
            y = conv_block(sd, "conv1.", x, act="relu6", q=q)
            y = conv_block(sd, "conv2.", y, stride=stride, padding=1, groups=y.shape[1], act="relu6", q=q)
            return conv_block(sd, "conv3.", y, act=None, q=q, residual=residual)
        if kind == "ResNeXtUnit":
            stride = kw["stride"]
            resize = (kw["in_channels"] != kw["out_channels"]) or (stride != 1)
            identity = conv_block(sd, "identity_conv.", x, stride=stride, act=None, q=q) if resize else x
            y = conv_block(sd, "body.conv1.", x, q=q)
            y = conv_block(sd, "body.conv2.", y, stride=stride, padding=1, groups=kw["cardinality"], q=q)
            return conv_block(sd, "body.conv3.", y, act=None, q=q, residual=identity, post_act="relu")
    raise NotImplementedError(kind)


# ---- branch / merge containers, interpolation, stand-alone BN + activation ------------------------------------
def f4_block_forward(name: str, sd: dict, x: torch.Tensor, quant: str | None = None) -> torch.Tensor:
    """The cases of tests/golden/cases.F4_CASES (structure: cases.build_f4_block) restated over a state_dict:
    Concurrent / SequentialConcurrent.forward (common/arch.py:87-99, 121-131), NormActivation.forward (common/tutti.py:188-191),
    InterpolationBlock.forward (tutti.py:232-246), channel_shuffle (tutti.py:267-291). 16-bit mode: every stored tensor is
    rounded where the GPU path stores it (each branch result / the merged sum once)."""
    q = Quant(quant)
    pool = lambda t: F.max_pool2d(t, kernel_size=3, stride=1, padding=1)       # noqa: E731 - nn.MaxPool2d(3, 1, 1) of the cases
    with torch.no_grad():
        x = q.r(x.float())
        if name == "norm_activation":
            return bn_act(sd, "bn.", x, q=q, act="relu")
        if name == "interp_bilinear_up2":
            return q.r(F.interpolate(x, size=(2 * x.shape[2], 2 * x.shape[3]), mode="bilinear", align_corners=True))
        if name == "interp_bilinear_noalign_size":
            return q.r(F.interpolate(x, size=(13, 10), mode="bilinear", align_corners=False))
        if name == "interp_nearest_up2":
            return q.r(F.interpolate(x, scale_factor=2, mode="nearest"))
        if name == "interp_nearest_ignores_up_false":       # tutti.py:240-246: scale_factor goes straight to F.interpolate, `up` unread
            return q.r(F.interpolate(x, scale_factor=2, mode="nearest"))
        if name == "interp_nearest_ignores_out_size":       # ... and so is `out_size`
            return q.r(F.interpolate(x, scale_factor=3, mode="nearest"))
        if name == "interp_bilinear_down2":
            return q.r(F.interpolate(x, size=(x.shape[2] // 2, x.shape[3] // 2), mode="bilinear", align_corners=True))
        if name == "concurrent_cat":
            b1 = conv_block(sd, "branch1.", x, q=q)
            b2 = conv_block(sd, "branch2.", x, padding=1, q=q)
            b3 = conv_block(sd, "branch3.conv2.", conv_block(sd, "branch3.conv1.", x, q=q), padding=1, q=q)
            return torch.cat((b1, b2, b3), dim=1)
        if name == "concurrent_cat_pool":
            b1 = conv_block(sd, "branch1.", x, padding=1, q=q)
            b2 = conv_block(sd, "branch2.conv.", pool(x), q=q)
            return torch.cat((b1, b2, pool(x)), dim=1)
        if name == "concurrent_sum":
            # GPU path: branch1 stored; branch2's convolution takes it as the epilogue residual (stored once); + pool, one add pass
            acc = conv_block(sd, "branch1.", x, q=q)
            acc = conv_block(sd, "branch2.", x, padding=1, q=q, residual=acc)
            return q.r(acc + pool(x))
        if name == "seq_concurrent":
            y1 = conv_block(sd, "conv1.", x, padding=1, q=q)
            y2 = conv_block(sd, "conv2.", y1, padding=1, q=q)
            return torch.cat((x, y1, y2), dim=1)
        if name == "channel_shuffle_g2":
            n, c, h, w = x.shape
            return x.view(n, 2, c // 2, h, w).transpose(1, 2).contiguous().view(n, c, h, w)
    raise KeyError(name)
