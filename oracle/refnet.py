"""
    ORACLE - test infrastructure, not product code.

    CPU restatement of the reference's conv-net inference path (osmr/pytorchcv 0.0.73) as pure
    functions over a `state_dict`. Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline`
    leg of `bench.py` may import this package; `pytorchcv_amd/` never does.

    Parity pin: the reference's own tests hold no tensor values ("parity
    unpinned" by the reference), so this restatement is pinned against outputs of the reference
    itself, imported in the build container by `tests/golden/make_golden.py`, and frozen as the
    fixtures under `tests/golden/` (`tests/test_oracle_golden.py` checks them, <= 1e-5).

    Every function cites the reference lines it follows (paths relative to the reference root).
    Arithmetic is delegated to the same third-party library the reference uses (torch, CPU, fp32);
    `oracle/cref.c` is an ATen-free restatement of the individual ops used to cross-check it.

    Two modes:
      * quant=None      - the reference semantics exactly: unfused conv -> batch_norm -> activation.
      * quant="bf16"/"fp16" - "quantisation-matched" restatement of the fused MI355X pipeline: weights
        and every tensor the GPU path stores to HBM are rounded to the 16-bit type at the same points,
        accumulation and the BN/activation/residual epilogue stay fp32.
"""

__all__ = ['Quant', 'conv_block', 'se_block', 'resnet_forward', 'mobilenetv2_forward', 'resnext_forward',
           'seresnet_forward', 'seresnext_forward', 'mobilenet_forward', 'mobilenetv3_forward', 'efficientnet_forward', 'preresnet_forward', 'densenet_forward', 'shufflenetv2_forward', 'vgg_forward', 'shuffle_unit', 'channel_shuffle', 'dense_unit', 'dense_transition', 'bn_act', 'pre_conv_chain', 'preres_unit', 'preres_init_block', 'tf_same_pad', 'effi_dws_unit', 'effi_inv_res_unit', 'forward', 'MODEL_ARCH', 'fold_bn', 'conv_then_se']

import math
import torch
import torch.nn.functional as F


class Quant(object):
    """Rounding points of the fused 16-bit pipeline (identity when dtype is None)."""
    def __init__(self, dtype: str | None):
        self.dtype = {None: None, "fp32": None, "bf16": torch.bfloat16, "fp16": torch.float16}[dtype]

    def r(self, t: torch.Tensor) -> torch.Tensor:
        if self.dtype is None:
            return t
        return t.to(self.dtype).to(torch.float32)

    # The individual rounding points (all `r` in the pipeline the GPU path runs). tests/tools/bf16_drift.py overrides them one
    # at a time to attribute the distance between the 16-bit forward and the fp32 reference forward.
    def rw(self, w: torch.Tensor) -> torch.Tensor:
        """convolution weights as MFMA operands"""
        return self.r(w)

    def rc(self, w: torch.Tensor) -> torch.Tensor:
        """classifier weights behind the global pool: the GPU path runs the head in fp32 (engine.FP32_HEAD)"""
        return w

    def ro(self, y: torch.Tensor, is_unit_output: bool = False) -> torch.Tensor:
        """an activation tensor stored to HBM (`is_unit_output`: the residual stream, written by a unit's last convolution)"""
        return self.r(y)

    def rp(self, f: torch.Tensor) -> torch.Tensor:
        """the pooled feature vector handed to the classifier, and activations inside the head: fp32 on the GPU path"""
        return f

    @property
    def on(self) -> bool:
        return self.dtype is not None


def _act(x: torch.Tensor, act: str | None) -> torch.Tensor:
    # pytorchcv/models/common/activ.py:50-81 (relu, relu6), :117-132 (sigmoid)
    if act is None:
        return x
    if act == "relu":
        return F.relu(x)
    if act == "relu6":
        return F.relu6(x)
    if act == "sigmoid":
        return torch.sigmoid(x)
    # activ.py:16-47: Swish x*sigmoid(x); HSigmoid relu6(x+3)/6; HSwish x*relu6(x+3)/6
    if act == "swish":
        return x * torch.sigmoid(x)
    if act == "hsigmoid":
        return F.relu6(x + 3.0) / 6.0
    if act == "hswish":
        return x * F.relu6(x + 3.0) / 6.0
    raise NotImplementedError(act)


def fold_bn(sd: dict, prefix: str, eps: float = 1e-5):
    """Eval-mode BatchNorm2d (common/norm.py:34-50) as per-channel scale/shift."""
    g = sd[prefix + "weight"].float()
    b = sd[prefix + "bias"].float()
    m = sd[prefix + "running_mean"].float()
    v = sd[prefix + "running_var"].float()
    scale = g / torch.sqrt(v + eps)
    shift = b - m * scale
    return scale, shift


def conv_block(sd: dict, prefix: str, x: torch.Tensor, stride=1, padding=0, dilation=1, groups=1,
               act: str | None = "relu", q: Quant | None = None, residual: torch.Tensor | None = None,
               post_act: str | None = None, eps: float = 1e-5, normalize: bool = True) -> torch.Tensor:
    """
    ConvBlock.forward, pytorchcv/models/common/conv.py:278-286: [ZeroPad2d] -> Conv2d -> [BN] -> [act].
    `residual`/`post_act` restate the add + activation that follows the block in the unit
    (resnet.py:227-228, mobilenetv2.py:69-70), which the GPU path fuses into the epilogue.
    """
    q = q or Quant(None)
    w = sd[prefix + "conv.weight"].float()
    bias = sd.get(prefix + "conv.bias", None)
    if isinstance(padding, (list, tuple)) and len(padding) == 4:
        # conv.py:245-249: 4-tuple padding is an explicit ZeroPad2d (left, right, top, bottom)
        x = F.pad(x, padding)
        padding = 0
    if q.on:
        y = F.conv2d(x, q.rw(w), None, stride, padding, dilation, groups)
        if normalize:
            scale, shift = fold_bn(sd, prefix + "bn.", eps)
            if bias is not None:
                shift = shift + bias.float() * scale
        else:
            scale = torch.ones(w.shape[0])
            shift = bias.float() if bias is not None else torch.zeros(w.shape[0])
        y = y * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    else:
        y = F.conv2d(x, w, bias, stride, padding, dilation, groups)
        if normalize:
            p = prefix + "bn."
            y = F.batch_norm(y, sd[p + "running_mean"], sd[p + "running_var"], sd[p + "weight"], sd[p + "bias"],
                             False, 0.0, eps)
    y = _act(y, act)
    if residual is not None:
        y = y + residual
    y = _act(y, post_act)
    return q.ro(y, residual is not None)


def se_block(sd: dict, prefix: str, x: torch.Tensor, q: Quant | None = None,
             residual: torch.Tensor | None = None, post_act: str | None = None,
             mid_act: str = "relu", out_act: str = "sigmoid") -> torch.Tensor:
    """
    SEBlock.forward, pytorchcv/models/common/att.py:94-105 (use_conv=True): AdaptiveAvgPool2d(1) ->
    1x1 conv + bias -> ReLU -> 1x1 conv + bias -> Sigmoid -> x * w.  The excitation runs in fp32 on
    the GPU path too (tiny), so only the stored result is rounded.
    """
    q = q or Quant(None)
    w = x.mean(dim=(2, 3), keepdim=True)
    w = F.conv2d(w, sd[prefix + "conv1.weight"], sd[prefix + "conv1.bias"])
    w = _act(w, mid_act)
    w = F.conv2d(w, sd[prefix + "conv2.weight"], sd[prefix + "conv2.bias"])
    w = _act(w, out_act)           # sigmoid, or hsigmoid in MobileNetV3 (mobilenetv3.py:64-69)
    y = x * w
    if residual is not None:
        y = y + residual
    y = _act(y, post_act)
    return q.r(y)


def conv_then_se(sd, conv_p, se_p, z, q, residual, post_act, eps=1e-5):
    """conv_block(act=None) -> SEBlock -> (+ residual) -> post_act (seresnet.py:63-72, seresnext.py:65-76). In fp32 this is the
    reference's arithmetic. In a 16-bit mode it follows the ROUNDING POINTS of the GPU pipeline, which runs the SE block inside
    the 1x1 convolution's launch (SEBlock.run_behind): BN(conv(.)) is affine, so the squeeze is BN(conv(mean_hw(z))) (fp32
    weights), the excitation runs first, and the convolution's unrounded fp32 result is scaled, added to the skip tensor,
    activated and only then rounded."""
    q = q or Quant(None)
    if not q.on:
        y = conv_block(sd, conv_p, z, act=None, q=q)
        return se_block(sd, se_p, y, q=q, residual=residual, post_act=post_act)
    w = sd[conv_p + "conv.weight"].float()
    scale, shift = fold_bn(sd, conv_p + "bn.", eps)
    bias = sd.get(conv_p + "conv.bias", None)
    if bias is not None:
        shift = shift + bias.float() * scale
    sq = F.conv2d(z.mean(dim=(2, 3), keepdim=True), w) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    g = F.conv2d(sq, sd[se_p + "conv1.weight"], sd[se_p + "conv1.bias"])
    g = torch.sigmoid(F.conv2d(F.relu(g), sd[se_p + "conv2.weight"], sd[se_p + "conv2.bias"]))
    y = F.conv2d(z, q.r(w)) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    y = y * g
    if residual is not None:
        y = y + residual
    return q.r(_act(y, post_act))


def _tap(taps, name, t):
    if taps is not None:
        taps[name] = t


def _res_init_block(sd, x, q):
    # ResInitBlock, resnet.py:232-263: conv7x7_block(stride 2, pad 3) + MaxPool2d(3, 2, 1)
    x = conv_block(sd, "features.init_block.conv.", x, stride=2, padding=3, q=q)
    return F.max_pool2d(x, kernel_size=3, stride=2, padding=1)


def _classifier(sd, x, q):
    # resnet.py:316-322,333-337: AvgPool2d(7, stride=1) -> view -> Linear
    x = q.rp(F.avg_pool2d(x, kernel_size=7, stride=1))
    x = x.view(x.size(0), -1)
    return F.linear(x, q.rc(sd["output.weight"].float()), sd["output.bias"].float())


def _res_layers(blocks: int, bottleneck: bool | None):
    # get_resnet, resnet.py:373-419
    if bottleneck is None:
        bottleneck = (blocks >= 50)
    table = {10: [1, 1, 1, 1], 12: [2, 1, 1, 1], 16: [2, 2, 2, 1], 18: [2, 2, 2, 2], 34: [3, 4, 6, 3],
             50: [3, 4, 6, 3], 101: [3, 4, 23, 3], 152: [3, 8, 36, 3], 200: [3, 24, 36, 3]}
    if blocks == 14:
        layers = [1, 1, 1, 1] if bottleneck else [2, 2, 1, 1]
    elif blocks == 26:
        layers = [2, 2, 2, 2] if bottleneck else [3, 3, 3, 3]
    elif blocks == 38 and bottleneck:
        layers = [3, 3, 3, 3]
    elif blocks in table:
        layers = table[blocks]
    else:
        raise ValueError("Unsupported ResNet with number of blocks: {}".format(blocks))
    cpl = [64, 128, 256, 512]
    if bottleneck:
        cpl = [c * 4 for c in cpl]
    channels = [[c] * n for c, n in zip(cpl, layers)]
    return channels, bottleneck


def _res_body(sd, p, x, stride, bottleneck, conv1_stride, q, residual, post_act):
    if bottleneck:
        # ResBottleneck, resnet.py:69-140
        y = conv_block(sd, p + "conv1.", x, stride=(stride if conv1_stride else 1), q=q)
        y = conv_block(sd, p + "conv2.", y, stride=(1 if conv1_stride else stride), padding=1, q=q)
        return conv_block(sd, p + "conv3.", y, act=None, q=q, residual=residual, post_act=post_act)
    # ResBlock, resnet.py:19-66
    y = conv_block(sd, p + "conv1.", x, stride=stride, padding=1, q=q)
    return conv_block(sd, p + "conv2.", y, padding=1, act=None, q=q, residual=residual, post_act=post_act)


def resnet_forward(sd, x, blocks, bottleneck=None, conv1_stride=True, q=None, taps=None, se=False):
    """ResNet.forward (resnet.py:333-337) / SEResNet.forward (seresnet.py:134-138) with ResUnit
    (resnet.py:221-229) / SEResUnit (seresnet.py:63-72)."""
    q = q or Quant(None)
    channels, bottleneck = _res_layers(blocks, bottleneck)
    x = q.r(x)
    x = _res_init_block(sd, x, q)
    _tap(taps, "init_block", x)
    in_ch = 64
    for i, cps in enumerate(channels):
        for j, out_ch in enumerate(cps):
            stride = 2 if (j == 0) and (i != 0) else 1
            p = "features.stage{}.unit{}.".format(i + 1, j + 1)
            if (in_ch != out_ch) or (stride != 1):
                identity = conv_block(sd, p + "identity_conv.", x, stride=stride, act=None, q=q)
            else:
                identity = x
            if se and bottleneck:
                y = conv_block(sd, p + "body.conv1.", x, stride=(stride if conv1_stride else 1), q=q)
                y = conv_block(sd, p + "body.conv2.", y, stride=(1 if conv1_stride else stride), padding=1, q=q)
                x = conv_then_se(sd, p + "body.conv3.", p + "se.", y, q, identity, "relu")
            elif se:
                y = _res_body(sd, p + "body.", x, stride, bottleneck, conv1_stride, q, None, None)
                x = se_block(sd, p + "se.", y, q=q, residual=identity, post_act="relu")
            else:
                x = _res_body(sd, p + "body.", x, stride, bottleneck, conv1_stride, q, identity, "relu")
            in_ch = out_ch
        _tap(taps, "stage{}".format(i + 1), x)
    return _classifier(sd, x, q)


def seresnet_forward(sd, x, blocks, q=None, taps=None):
    return resnet_forward(sd, x, blocks, q=q, taps=taps, se=True)


def resnext_forward(sd, x, blocks, cardinality, bottleneck_width, q=None, taps=None):
    """ResNeXt.forward (resnext.py:186-190); ResNeXtUnit (resnext.py:108-116); ResNeXtBottleneck
    (resnext.py:41-65): stride sits on the grouped 3x3."""
    q = q or Quant(None)
    layers = {14: [1, 1, 1, 1], 26: [2, 2, 2, 2], 38: [3, 3, 3, 3], 50: [3, 4, 6, 3], 101: [3, 4, 23, 3]}[blocks]
    channels = [[c] * n for c, n in zip([256, 512, 1024, 2048], layers)]
    x = q.r(x)
    x = _res_init_block(sd, x, q)
    _tap(taps, "init_block", x)
    in_ch = 64
    for i, cps in enumerate(channels):
        for j, out_ch in enumerate(cps):
            stride = 2 if (j == 0) and (i != 0) else 1
            p = "features.stage{}.unit{}.".format(i + 1, j + 1)
            if (in_ch != out_ch) or (stride != 1):
                identity = conv_block(sd, p + "identity_conv.", x, stride=stride, act=None, q=q)
            else:
                identity = x
            y = conv_block(sd, p + "body.conv1.", x, q=q)
            y = conv_block(sd, p + "body.conv2.", y, stride=stride, padding=1, groups=cardinality, q=q)
            x = conv_block(sd, p + "body.conv3.", y, act=None, q=q, residual=identity, post_act="relu")
            in_ch = out_ch
        _tap(taps, "stage{}".format(i + 1), x)
    return _classifier(sd, x, q)


def seresnext_forward(sd, x, blocks, cardinality, bottleneck_width, q=None, taps=None):
    """SEResNeXt.forward / SEResNeXtUnit (reference seresnext.py:17-76,147-151): ResNeXt body + SEBlock before the skip add."""
    q = q or Quant(None)
    layers = {50: [3, 4, 6, 3], 101: [3, 4, 23, 3]}[blocks]
    channels = [[c] * n for c, n in zip([256, 512, 1024, 2048], layers)]
    x = q.r(x)
    x = _res_init_block(sd, x, q)
    _tap(taps, "init_block", x)
    in_ch = 64
    for i, cps in enumerate(channels):
        for j, out_ch in enumerate(cps):
            stride = 2 if (j == 0) and (i != 0) else 1
            p = "features.stage{}.unit{}.".format(i + 1, j + 1)
            if (in_ch != out_ch) or (stride != 1):
                identity = conv_block(sd, p + "identity_conv.", x, stride=stride, act=None, q=q)
            else:
                identity = x
            y = conv_block(sd, p + "body.conv1.", x, q=q)
            y = conv_block(sd, p + "body.conv2.", y, stride=stride, padding=1, groups=cardinality, q=q)
            x = conv_then_se(sd, p + "body.conv3.", p + "se.", y, q, identity, "relu")
            in_ch = out_ch
        _tap(taps, "stage{}".format(i + 1), x)
    return _classifier(sd, x, q)


def mobilenet_forward(sd, x, width_scale=1.0, q=None, taps=None):
    """MobileNet.forward (reference mobilenet.py:92-96): 3x3/2 stem, depthwise-separable units (DwsConvBlock,
    common/conv.py:546-618: depthwise ConvBlock then pointwise ConvBlock), AvgPool2d(7), Linear."""
    q = q or Quant(None)
    channels = [[32], [64], [128, 128], [256, 256], [512] * 6, [1024, 1024]]
    if width_scale != 1.0:
        channels = [[int(c * width_scale) for c in ci] for ci in channels]
    x = q.r(x)
    x = conv_block(sd, "features.init_block.", x, stride=2, padding=1, q=q)
    _tap(taps, "init_block", x)
    for i, cps in enumerate(channels[1:]):
        for j, out_ch in enumerate(cps):
            stride = 2 if (j == 0) and (i != 0) else 1
            p = "features.stage{}.unit{}.".format(i + 1, j + 1)
            x = conv_block(sd, p + "dw_conv.", x, stride=stride, padding=1, groups=x.shape[1], q=q)
            x = conv_block(sd, p + "pw_conv.", x, q=q)
        _tap(taps, "stage{}".format(i + 1), x)
    return _classifier(sd, x, q)


def mobilenetv2_forward(sd, x, width_scale=1.0, q=None, taps=None):
    """MobileNetV2.forward (mobilenetv2.py:152-156); LinearBottleneck (mobilenetv2.py:62-71);
    channel plan of get_mobilenetv2 (mobilenetv2.py:183-203)."""
    q = q or Quant(None)
    layers = [1, 2, 3, 4, 3, 3, 1]
    downsample = [0, 1, 1, 1, 0, 1, 0]
    cpl = [16, 24, 32, 64, 96, 160, 320]
    channels = [[]]
    for c, n, d in zip(cpl, layers, downsample):
        if d != 0:
            channels = channels + [[c] * n]
        else:
            channels = channels[:-1] + [channels[-1] + [c] * n]
    init_ch, final_ch = 32, 1280
    if width_scale != 1.0:
        channels = [[int(c * width_scale) for c in ci] for ci in channels]
        init_ch = int(init_ch * width_scale)
        if width_scale > 1.0:
            final_ch = int(final_ch * width_scale)
    x = q.r(x)
    x = conv_block(sd, "features.init_block.", x, stride=2, padding=1, act="relu6", q=q)
    _tap(taps, "init_block", x)
    in_ch = init_ch
    for i, cps in enumerate(channels):
        for j, out_ch in enumerate(cps):
            stride = 2 if (j == 0) and (i != 0) else 1
            p = "features.stage{}.unit{}.".format(i + 1, j + 1)
            residual = x if (in_ch == out_ch and stride == 1) else None
            y = conv_block(sd, p + "conv1.", x, act="relu6", q=q)     # expansion (always present for _w1)
            mid = y.shape[1]
            y = conv_block(sd, p + "conv2.", y, stride=stride, padding=1, groups=mid, act="relu6", q=q)
            x = conv_block(sd, p + "conv3.", y, act=None, q=q, residual=residual)
            in_ch = out_ch
        _tap(taps, "stage{}".format(i + 1), x)
    x = conv_block(sd, "features.final_block.", x, act="relu6", q=q)
    x = q.rp(F.avg_pool2d(x, kernel_size=7, stride=1))
    # mobilenetv2.py:138-141,154-155: bias-free 1x1 conv classifier, then view
    x = F.conv2d(x, q.rc(sd["output.weight"].float()))
    return x.view(x.size(0), -1)


def mobilenetv3_forward(sd, x, version="large", q=None, taps=None):
    """MobileNetV3.forward (mobilenetv3.py:277-281), MobileNetV3Unit.forward (:82-93), MobileNetV3FinalBlock (:127-131),
    MobileNetV3Classifier (:167-174); per-unit activation / stride tables of get_mobilenetv3 (:311-332). Channel counts,
    kernel sizes, the presence of the expansion conv and of SE are read from the state_dict itself."""
    q = q or Quant(None)
    if version == "small":
        use_relu = [[1], [1, 1], [0, 0, 0, 0, 0], [0, 0, 0]]
        first_stride = True
    elif version == "large":
        use_relu = [[1], [1, 1], [1, 1, 1], [0, 0, 0, 0, 0, 0], [0, 0, 0]]
        first_stride = False
    else:
        raise ValueError(version)
    x = q.r(x)
    x = conv_block(sd, "features.init_block.", x, stride=2, padding=1, act="hswish", q=q)
    _tap(taps, "init_block", x)
    for i, relu_flags in enumerate(use_relu):
        for j, relu_flag in enumerate(relu_flags):
            stride = 2 if (j == 0) and ((i != 0) or first_stride) else 1
            act = "relu" if relu_flag == 1 else "hswish"
            p = "features.stage{}.unit{}.".format(i + 1, j + 1)
            out_ch = sd[p + "conv2.conv.weight"].shape[0]
            residual = x if (x.shape[1] == out_ch and stride == 1) else None
            y = x
            if (p + "exp_conv.conv.weight") in sd:
                y = conv_block(sd, p + "exp_conv.", y, act=act, q=q)
            wdw = sd[p + "conv1.conv.weight"]
            k = wdw.shape[-1]
            y = conv_block(sd, p + "conv1.", y, stride=stride, padding=k // 2, groups=wdw.shape[0], act=act, q=q)
            if (p + "se.conv1.weight") in sd:
                y = se_block(sd, p + "se.", y, q=q, out_act="hsigmoid")
            x = conv_block(sd, p + "conv2.", y, act=None, q=q, residual=residual)
        _tap(taps, "stage{}".format(i + 1), x)
    x = conv_block(sd, "features.final_block.conv.", x, act="hswish", q=q)
    if "features.final_block.se.conv1.weight" in sd:
        x = se_block(sd, "features.final_block.se.", x, q=q, out_act="hsigmoid")
    x = q.rp(F.avg_pool2d(x, kernel_size=7, stride=1))
    x = q.rp(_act(F.conv2d(x, q.rc(sd["output.conv1.weight"].float())), "hswish"))
    x = F.conv2d(x, q.rc(sd["output.conv2.weight"].float()), sd["output.conv2.bias"].float())   # dropout: identity in eval
    return x.view(x.size(0), -1)


def bn_act(sd, prefix, x, q=None, act="relu", eps=1e-5):
    """BatchNorm2d(eval) + activation as a stand-alone pass (PreConvBlock's front half, conv.py:776-779; PreResActivation,
    preresnet.py:219-222): one rounding point when the GPU path runs it as its own launch."""
    q = q or Quant(None)
    if q.on:
        scale, shift = fold_bn(sd, prefix, eps)
        y = x * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    else:
        y = F.batch_norm(x, sd[prefix + "running_mean"], sd[prefix + "running_var"], sd[prefix + "weight"],
                         sd[prefix + "bias"], False, 0.0, eps)
    return q.r(_act(y, act))


def pre_conv_chain(sd, prefixes, strides, a, q=None, residual=None, eps=1e-5, se_prefix=None):
    """A run of PreConvBlocks (conv.py:776-786) whose first pre-activation has already been applied to `a`: block i's
    Conv2d, then block i+1's BatchNorm + ReLU. The GPU path fuses exactly that pair into one launch, so in the
    quantisation-matched mode the value is rounded once per pair; the skip add joins the last convolution."""
    q = q or Quant(None)
    for i, p in enumerate(prefixes):
        w = sd[p + "conv.weight"].float()
        b = sd.get(p + "conv.bias", None)
        y = F.conv2d(a, q.r(w), b.float() if b is not None else None, strides[i], w.shape[-1] // 2)
        if i + 1 < len(prefixes):
            nb = prefixes[i + 1] + "bn."
            if q.on:
                scale, shift = fold_bn(sd, nb, eps)
                y = y * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
            else:
                y = F.batch_norm(y, sd[nb + "running_mean"], sd[nb + "running_var"], sd[nb + "weight"], sd[nb + "bias"],
                                 False, 0.0, eps)
            a = q.r(F.relu(y))
        else:
            if se_prefix is not None:
                # 16-bit pipeline (SEBlock.run_behind): the last convolution is linear in its pre-activated input, so the squeeze
                # is conv(mean_hw(a)) with fp32 weights, and its unrounded result is scaled, joined with the skip, then rounded
                sq = F.conv2d(a.mean(dim=(2, 3), keepdim=True), w, b.float() if b is not None else None)
                g = F.conv2d(sq, sd[se_prefix + "conv1.weight"], sd[se_prefix + "conv1.bias"])
                g = torch.sigmoid(F.conv2d(F.relu(g), sd[se_prefix + "conv2.weight"], sd[se_prefix + "conv2.bias"]))
                y = y * g
            if residual is not None:
                y = y + residual
            a = q.r(y)
    return a


def preres_unit(sd, p, x, stride, bottleneck, conv1_stride, q=None):
    """PreResUnit.forward (preresnet.py:157-164) over PreResBlock (:58-61) / PreResBottleneck (:102-106); with an
    `se.` sub-block in the state_dict it is SEPreResUnit.forward (sepreresnet.py:63-71): body -> SE -> + identity."""
    q = q or Quant(None)
    if bottleneck:
        names = ["conv1.", "conv2.", "conv3."]
        strides = [stride if conv1_stride else 1, 1 if conv1_stride else stride, 1]
    else:
        names = ["conv1.", "conv2."]
        strides = [stride, 1]
    pre = bn_act(sd, p + "body.conv1.bn.", x, q)
    identity = x
    if (p + "identity_conv.weight") in sd:
        b = sd.get(p + "identity_conv.bias", None)
        identity = q.r(F.conv2d(pre, q.r(sd[p + "identity_conv.weight"].float()), b.float() if b is not None else None, stride))
    if (p + "se.conv1.weight") in sd:
        if q.on and bottleneck:      # the GPU pipeline runs the SE block inside the last 1x1 convolution: its rounding points
            return pre_conv_chain(sd, [p + "body." + n for n in names], strides, pre, q, residual=identity, se_prefix=p + "se.")
        y = pre_conv_chain(sd, [p + "body." + n for n in names], strides, pre, q)
        return se_block(sd, p + "se.", y, q=q, residual=identity)
    return pre_conv_chain(sd, [p + "body." + n for n in names], strides, pre, q, residual=identity)


def preres_init_block(sd, p, x, q=None):
    """PreResInitBlock.forward (preresnet.py:190-196): conv7x7/2 -> BN -> ReLU -> MaxPool(3,2,1); conv/bn live on the block."""
    q = q or Quant(None)
    w = sd[p + "conv.weight"].float()
    y = F.conv2d(x, q.r(w), None, 2, 3)
    if q.on:
        scale, shift = fold_bn(sd, p + "bn.")
        y = y * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    else:
        y = F.batch_norm(y, sd[p + "bn.running_mean"], sd[p + "bn.running_var"], sd[p + "bn.weight"], sd[p + "bn.bias"],
                         False, 0.0, 1e-5)
    return F.max_pool2d(q.r(F.relu(y)), kernel_size=3, stride=2, padding=1)


def preresnet_forward(sd, x, blocks=18, bottleneck=None, conv1_stride=True, q=None, taps=None):
    """PreResNet.forward (preresnet.py:279-283); stage plan of get_preresnet (:320-369): the unit count per stage is read
    from the state_dict."""
    q = q or Quant(None)
    if bottleneck is None:
        bottleneck = blocks >= 50
    x = preres_init_block(sd, "features.init_block.", q.r(x), q)
    _tap(taps, "init_block", x)
    for i in range(4):
        j = 0
        while ("features.stage{}.unit{}.body.conv1.conv.weight".format(i + 1, j + 1)) in sd:
            stride = 1 if (i == 0) or (j != 0) else 2
            x = preres_unit(sd, "features.stage{}.unit{}.".format(i + 1, j + 1), x, stride, bottleneck, conv1_stride, q)
            j += 1
        _tap(taps, "stage{}".format(i + 1), x)
    x = bn_act(sd, "features.post_activ.bn.", x, q)
    return _classifier(sd, x, q)


def dense_unit(sd, p, x, q=None):
    """DenseUnit.forward (densenet.py:51-59): BN-ReLU-1x1, BN-ReLU-3x3, torch.cat((identity, x), dim=1)."""
    q = q or Quant(None)
    pre = bn_act(sd, p + "conv1.bn.", x, q)
    y = pre_conv_chain(sd, [p + "conv1.", p + "conv2."], [1, 1], pre, q)
    return torch.cat((x, y), dim=1)


def dense_transition(sd, p, x, q=None):
    """TransitionBlock.forward (densenet.py:87-91): pre-activated 1x1 convolution, AvgPool2d(2, 2)."""
    q = q or Quant(None)
    y = pre_conv_chain(sd, [p + "conv."], [1], bn_act(sd, p + "conv.bn.", x, q), q)
    return q.r(F.avg_pool2d(y, kernel_size=2, stride=2))


def densenet_forward(sd, x, q=None, taps=None):
    """DenseNet.forward (densenet.py:156-160); stages/units are read from the state_dict (get_densenet, :204-235)."""
    q = q or Quant(None)
    x = preres_init_block(sd, "features.init_block.", q.r(x), q)
    _tap(taps, "init_block", x)
    i = 0
    while ("features.stage{}.unit1.conv1.conv.weight".format(i + 1)) in sd:
        sp = "features.stage{}.".format(i + 1)
        if (sp + "trans{}.conv.conv.weight".format(i + 1)) in sd:
            x = dense_transition(sd, sp + "trans{}.".format(i + 1), x, q)
        j = 0
        while (sp + "unit{}.conv1.conv.weight".format(j + 1)) in sd:
            x = dense_unit(sd, sp + "unit{}.".format(j + 1), x, q)
            j += 1
        _tap(taps, "stage{}".format(i + 1), x)
        i += 1
    x = bn_act(sd, "features.post_activ.bn.", x, q)
    return _classifier(sd, x, q)


def _conv_bn(sd, conv_p, bn_p, x, q, stride=1, padding=0, groups=1, act=None, eps=1e-5):
    """Bare Conv2d followed by a stand-alone BatchNorm2d (+ activation), as ShuffleUnit spells it (shufflenetv2.py:71-88); the GPU
    path fuses the three, so the quantised mode rounds once."""
    w = sd[conv_p + "weight"].float()
    if q.on:
        y = F.conv2d(x, q.r(w), None, stride, padding, 1, groups)
        scale, shift = fold_bn(sd, bn_p, eps)
        y = y * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    else:
        y = F.conv2d(x, w, None, stride, padding, 1, groups)
        y = F.batch_norm(y, sd[bn_p + "running_mean"], sd[bn_p + "running_var"], sd[bn_p + "weight"], sd[bn_p + "bias"],
                         False, 0.0, eps)
    return q.r(_act(y, act))


def channel_shuffle(x, groups):
    """common/tutti.py:267-291."""
    b, c, h, w = x.shape
    return x.view(b, groups, c // groups, h, w).transpose(1, 2).contiguous().view(b, c, h, w)


def shuffle_unit(sd, p, x, downsample, q=None):
    """ShuffleUnit.forward (shufflenetv2.py:69-91), use_se / use_residual False as in the registry models."""
    q = q or Quant(None)
    if downsample:
        y1 = _conv_bn(sd, p + "dw_conv4.", p + "dw_bn4.", x, q, stride=2, padding=1, groups=x.shape[1])
        y1 = _conv_bn(sd, p + "expand_conv5.", p + "expand_bn5.", y1, q, act="relu")
        x2 = x
    else:
        y1, x2 = torch.chunk(x, chunks=2, dim=1)
    y2 = _conv_bn(sd, p + "compress_conv1.", p + "compress_bn1.", x2, q, act="relu")
    y2 = _conv_bn(sd, p + "dw_conv2.", p + "dw_bn2.", y2, q, stride=(2 if downsample else 1), padding=1, groups=y2.shape[1])
    y2 = _conv_bn(sd, p + "expand_conv3.", p + "expand_bn3.", y2, q, act="relu")
    return channel_shuffle(torch.cat((y1, y2), dim=1), 2)


def shufflenetv2_forward(sd, x, q=None, taps=None):
    """ShuffleNetV2.forward (shufflenetv2.py:176-180); ShuffleInitBlock (:117-120) with MaxPool2d(3, 2, 0, ceil_mode=True)."""
    q = q or Quant(None)
    x = conv_block(sd, "features.init_block.conv.", q.r(x), stride=2, padding=1, q=q)
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=0, ceil_mode=True)
    _tap(taps, "init_block", x)
    i = 0
    while ("features.stage{}.unit1.compress_conv1.weight".format(i + 1)) in sd:
        j = 0
        while ("features.stage{}.unit{}.compress_conv1.weight".format(i + 1, j + 1)) in sd:
            x = shuffle_unit(sd, "features.stage{}.unit{}.".format(i + 1, j + 1), x, downsample=(j == 0), q=q)
            j += 1
        _tap(taps, "stage{}".format(i + 1), x)
        i += 1
    x = conv_block(sd, "features.final_block.", x, q=q)
    return _classifier(sd, x, q)


def vgg_forward(sd, x, blocks=11, use_bn=False, q=None, taps=None):
    """VGG.forward, pytorchcv/models/vgg.py:80-145: per stage conv3x3_block(bias, [BN], ReLU) x n -> MaxPool2d(2, 2);
    view(N, -1) in NCHW order -> VGGOutputBlock (vgg.py:45-77): fc1 + ReLU -> fc2 + ReLU -> fc3 (Dropout = identity in eval).
    The 16-bit pipeline rounds each stored activation and the weights; logits stay fp32."""
    q = q or Quant(None)
    layers = {11: [1, 1, 2, 2, 2], 13: [2, 2, 2, 2, 2], 16: [2, 2, 3, 3, 3], 19: [2, 2, 4, 4, 4]}[blocks]   # vgg.py:164-173
    x = q.r(x)
    for i, n in enumerate(layers):
        for j in range(n):
            x = conv_block(sd, "features.stage{}.unit{}.".format(i + 1, j + 1), x, padding=1, q=q, normalize=use_bn)
        x = F.max_pool2d(x, kernel_size=2, stride=2, padding=0)
        _tap(taps, "stage{}".format(i + 1), x)
    x = x.reshape(x.size(0), -1)
    for name in ("fc1.fc", "fc2.fc"):
        x = q.r(F.relu(F.linear(x, q.r(sd["output." + name + ".weight"].float()), sd["output." + name + ".bias"].float())))
    return F.linear(x, q.r(sd["output.fc3.weight"].float()), sd["output.fc3.bias"].float())


def tf_same_pad(h, w, kernel_size, stride=1, dilation=1):
    """calc_tf_padding, efficientnet.py:27-55. Returned in F.pad order: the reference hands (pad_h//2, pad_h - pad_h//2,
    pad_w//2, pad_w - pad_w//2) to F.pad, which reads it as (left, right, top, bottom)."""
    oh = math.ceil(float(h) / stride)
    ow = math.ceil(float(w) / stride)
    pad_h = max((oh - 1) * stride + (kernel_size - 1) * dilation + 1 - h, 0)
    pad_w = max((ow - 1) * stride + (kernel_size - 1) * dilation + 1 - w, 0)
    return (pad_h // 2, pad_h - pad_h // 2, pad_w // 2, pad_w - pad_w // 2)


def effi_dws_unit(sd, p, x, q, tf_mode, eps, residual_ok=True):
    """EffiDwsConvUnit.forward, efficientnet.py:105-115 (the depthwise conv always has stride 1)."""
    out_ch = sd[p + "pw_conv.conv.weight"].shape[0]
    residual = x if (residual_ok and x.shape[1] == out_ch) else None
    pad = tf_same_pad(x.shape[2], x.shape[3], 3) if tf_mode else 1
    y = conv_block(sd, p + "dw_conv.", x, padding=pad, groups=x.shape[1], act="swish", q=q, eps=eps)
    y = se_block(sd, p + "se.", y, q=q, mid_act="swish")
    return conv_block(sd, p + "pw_conv.", y, act=None, q=q, residual=residual, eps=eps)


def effi_inv_res_unit(sd, p, x, q, stride, tf_mode, eps):
    """EffiInvResUnit.forward, efficientnet.py:185-197."""
    out_ch = sd[p + "conv3.conv.weight"].shape[0]
    residual = x if (x.shape[1] == out_ch and stride == 1) else None
    y = conv_block(sd, p + "conv1.", x, act="swish", q=q, eps=eps)
    wdw = sd[p + "conv2.conv.weight"]
    k = wdw.shape[-1]
    pad = tf_same_pad(y.shape[2], y.shape[3], k, stride) if tf_mode else k // 2
    y = conv_block(sd, p + "conv2.", y, stride=stride, padding=pad, groups=wdw.shape[0], act="swish", q=q, eps=eps)
    if (p + "se.conv1.weight") in sd:
        y = se_block(sd, p + "se.", y, q=q, mid_act="swish")
    return conv_block(sd, p + "conv3.", y, act=None, q=q, residual=residual, eps=eps)


def efficientnet_forward(sd, x, version="b0", tf_mode=False, bn_eps=1e-5, q=None, taps=None):
    """EfficientNet.forward (efficientnet.py:354-358); stage/stride plan of get_efficientnet (:441-466) - the number of
    units per stage is read from the state_dict, the per-stage strides are those of the reference table."""
    q = q or Quant(None)
    strides = [1, 2, 2, 2, 2]              # strides_per_stage after merging the non-downsampling layers (:448,463-465)
    x = q.r(x)
    pad = tf_same_pad(x.shape[2], x.shape[3], 3, 2) if tf_mode else 1
    x = conv_block(sd, "features.init_block.conv.", x, stride=2, padding=pad, act="swish", q=q, eps=bn_eps)
    _tap(taps, "init_block", x)
    for i, stage_stride in enumerate(strides):
        j = 0
        while True:
            p = "features.stage{}.unit{}.".format(i + 1, j + 1)
            if i == 0:
                if (p + "dw_conv.conv.weight") not in sd:
                    break
                x = effi_dws_unit(sd, p, x, q, tf_mode, bn_eps, residual_ok=(stage_stride == 1 or j > 0))
            else:
                if (p + "conv1.conv.weight") not in sd:
                    break
                x = effi_inv_res_unit(sd, p, x, q, stage_stride if j == 0 else 1, tf_mode, bn_eps)
            j += 1
        _tap(taps, "stage{}".format(i + 1), x)
    x = conv_block(sd, "features.final_block.", x, act="swish", q=q, eps=bn_eps)
    x = q.rp(x.mean(dim=(2, 3), keepdim=True))                             # AdaptiveAvgPool2d(1), :339
    x = x.view(x.size(0), -1)
    return F.linear(x, q.rc(sd["output.fc.weight"].float()), sd["output.fc.bias"].float())   # dropout: identity in eval


MODEL_ARCH = {
    "resnet18": ("resnet", dict(blocks=18)),
    "resnet34": ("resnet", dict(blocks=34)),
    "resnet50": ("resnet", dict(blocks=50)),
    "resnet50b": ("resnet", dict(blocks=50, conv1_stride=False)),
    "resnet101": ("resnet", dict(blocks=101)),
    "resnet152": ("resnet", dict(blocks=152)),
    "resnet10": ("resnet", dict(blocks=10)), "resnet12": ("resnet", dict(blocks=12)), "resnet14": ("resnet", dict(blocks=14)),
    "resnet16": ("resnet", dict(blocks=16)), "resnet26": ("resnet", dict(blocks=26, bottleneck=False)),
    "resnetbc14b": ("resnet", dict(blocks=14, bottleneck=True, conv1_stride=False)),
    "resnetbc26b": ("resnet", dict(blocks=26, bottleneck=True, conv1_stride=False)),
    "resnetbc38b": ("resnet", dict(blocks=38, bottleneck=True, conv1_stride=False)),
    "resnet101b": ("resnet", dict(blocks=101, conv1_stride=False)), "resnet152b": ("resnet", dict(blocks=152, conv1_stride=False)),
    "mobilenetv2_w1": ("mobilenetv2", dict(width_scale=1.0)),
    "mobilenetv2_w3d4": ("mobilenetv2", dict(width_scale=0.75)),
    "mobilenetv2_wd2": ("mobilenetv2", dict(width_scale=0.5)),
    "mobilenetv2_wd4": ("mobilenetv2", dict(width_scale=0.25)),
    "resnext50_32x4d": ("resnext", dict(blocks=50, cardinality=32, bottleneck_width=4)),
    "resnext101_32x4d": ("resnext", dict(blocks=101, cardinality=32, bottleneck_width=4)),
    "resnext101_64x4d": ("resnext", dict(blocks=101, cardinality=64, bottleneck_width=4)),
    "seresnet18": ("seresnet", dict(blocks=18)),
    "seresnet50": ("seresnet", dict(blocks=50)),
    "seresnet101": ("seresnet", dict(blocks=101)),
    "seresnext50_32x4d": ("seresnext", dict(blocks=50, cardinality=32, bottleneck_width=4)),
    "seresnext101_32x4d": ("seresnext", dict(blocks=101, cardinality=32, bottleneck_width=4)),
    "mobilenet_w1": ("mobilenet", dict(width_scale=1.0)),
    "mobilenet_wd2": ("mobilenet", dict(width_scale=0.5)),
    "mobilenet_w3d4": ("mobilenet", dict(width_scale=0.75)), "mobilenet_wd4": ("mobilenet", dict(width_scale=0.25)),
}
for _v in ("small", "large"):
    for _t in ("w7d20", "wd2", "w3d4", "w1", "w5d4"):
        MODEL_ARCH["mobilenetv3_{}_{}".format(_v, _t)] = ("mobilenetv3", dict(version=_v))
for _n, _kw in {"preresnet10": dict(blocks=10), "preresnet12": dict(blocks=12), "preresnet14": dict(blocks=14),
                 "preresnetbc14b": dict(blocks=14, bottleneck=True, conv1_stride=False), "preresnet16": dict(blocks=16),
                 "preresnet18": dict(blocks=18), "preresnet26": dict(blocks=26, bottleneck=False),
                 "preresnetbc26b": dict(blocks=26, bottleneck=True, conv1_stride=False), "preresnet34": dict(blocks=34),
                 "preresnetbc38b": dict(blocks=38, bottleneck=True, conv1_stride=False), "preresnet50": dict(blocks=50),
                 "preresnet50b": dict(blocks=50, conv1_stride=False), "preresnet101": dict(blocks=101),
                 "preresnet101b": dict(blocks=101, conv1_stride=False), "preresnet152": dict(blocks=152),
                 "preresnet152b": dict(blocks=152, conv1_stride=False), "preresnet200": dict(blocks=200),
                 "preresnet200b": dict(blocks=200, conv1_stride=False),
                 "preresnet269b": dict(blocks=269, conv1_stride=False)}.items():
    MODEL_ARCH[_n] = ("preresnet", _kw)
    if _n != "preresnet269b":
        MODEL_ARCH["se" + _n] = ("preresnet", _kw)          # SE-PreResNet: same trunk, `se.` blocks in the state_dict
for _n in ("shufflenetv2_wd2", "shufflenetv2_w1", "shufflenetv2_w3d2", "shufflenetv2_w2"):
    MODEL_ARCH[_n] = ("shufflenetv2", dict())
for _b in (11, 13, 16, 19):
    MODEL_ARCH["vgg{}".format(_b)] = ("vgg", dict(blocks=_b))
    MODEL_ARCH["bn_vgg{}".format(_b)] = ("vgg", dict(blocks=_b, use_bn=True))
    MODEL_ARCH["bn_vgg{}b".format(_b)] = ("vgg", dict(blocks=_b, use_bn=True))
for _n in ("densenet121", "densenet161", "densenet169", "densenet201"):
    MODEL_ARCH[_n] = ("densenet", dict())
for _v in ("b0", "b1", "b2", "b3", "b4", "b5", "b6", "b7", "b8"):
    MODEL_ARCH["efficientnet_" + _v] = ("efficientnet", dict(version=_v))
    for _t in ("b", "c"):
        MODEL_ARCH["efficientnet_" + _v + _t] = ("efficientnet", dict(version=_v, tf_mode=True, bn_eps=1e-3))


_FAMILY = {"resnet": resnet_forward, "mobilenetv2": mobilenetv2_forward, "resnext": resnext_forward,
           "seresnet": seresnet_forward, "seresnext": seresnext_forward, "mobilenet": mobilenet_forward,
           "mobilenetv3": mobilenetv3_forward, "efficientnet": efficientnet_forward,
           "preresnet": preresnet_forward, "densenet": densenet_forward,
           "shufflenetv2": shufflenetv2_forward, "vgg": vgg_forward}


def forward(model_name: str, sd: dict, x: torch.Tensor, quant: str | None = None, taps: dict | None = None):
    """CPU forward of `model_name` (`x`: NCHW fp32 -> [N, num_classes] fp32)."""
    family, kw = MODEL_ARCH[model_name]
    with torch.no_grad():
        return _FAMILY[family](sd, x.float(), q=Quant(quant), taps=taps, **kw)
