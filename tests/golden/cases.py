"""
    Block-level parity cases shared by the fixture generator (tests/golden/make_golden.py, which builds them
    from the imported reference) and by the tests (which build the same blocks from pytorchcv_amd / the oracle).
    Each case: name, kind (constructor in the reference), kwargs, input shape. Data only.
"""

MODELS = ["resnet18", "resnet50", "mobilenetv2_w1", "resnext101_32x4d", "seresnet50", "seresnext50_32x4d", "mobilenet_w1",
          "mobilenetv3_large_w1", "mobilenetv3_small_w1", "efficientnet_b0", "efficientnet_b0b",
          "preresnet18", "preresnet50", "sepreresnet18", "densenet121", "shufflenetv2_w1", "vgg11", "bn_vgg11b"]

# known answers from the reference's own asserts (resnet.py:983,988; mobilenetv2.py:436; resnext.py:547;
# model_metainfos.csv:81,317,322,347,355) and SURVEY.md section 8a16 (state_dict key counts)
PARAM_COUNTS = {"resnet18": 11689512, "resnet50": 25557032, "mobilenetv2_w1": 3504960,
                "resnext101_32x4d": 44177704, "seresnet50": 28088024, "seresnext50_32x4d": 27559896,
                "mobilenet_w1": 4231976, "mobilenetv3_large_w1": 5481752, "mobilenetv3_small_w1": 2945288,
                "efficientnet_b0": 5288548, "efficientnet_b0b": 5288548, "preresnet18": 11687848, "preresnet50": 25549480,
                "sepreresnet18": 11776928, "densenet121": 7978856, "shufflenetv2_w1": 2278604,
                "vgg11": 132863336, "bn_vgg11b": 132868840}
KEY_COUNTS = {"resnet18": 122, "resnet50": 320, "mobilenetv2_w1": 319, "resnext101_32x4d": 626}

BLOCK_CASES = [
    dict(name="conv1x1_relu", kind="conv1x1_block", kwargs=dict(in_channels=32, out_channels=64), x=(2, 32, 10, 10)),
    dict(name="conv1x1_s2_noact", kind="conv1x1_block",
         kwargs=dict(in_channels=64, out_channels=128, stride=2, activation=None), x=(2, 64, 14, 14)),
    dict(name="conv1x1_c24_c144_relu6", kind="conv1x1_block",
         kwargs=dict(in_channels=24, out_channels=144, activation="relu6"), x=(2, 24, 9, 9)),
    dict(name="conv1x1_c144_c24_noact", kind="conv1x1_block",
         kwargs=dict(in_channels=144, out_channels=24, activation=None), x=(2, 144, 9, 9)),
    dict(name="conv3x3_s1", kind="conv3x3_block", kwargs=dict(in_channels=64, out_channels=64), x=(2, 64, 14, 14)),
    dict(name="conv3x3_s2", kind="conv3x3_block",
         kwargs=dict(in_channels=64, out_channels=128, stride=2), x=(2, 64, 15, 15)),
    dict(name="conv3x3_c128_7x7", kind="conv3x3_block", kwargs=dict(in_channels=128, out_channels=128), x=(3, 128, 7, 7)),
    dict(name="conv3x3_dil2", kind="conv3x3_block",
         kwargs=dict(in_channels=32, out_channels=32, padding=2, dilation=2), x=(1, 32, 12, 12)),
    dict(name="conv7x7_stem", kind="conv7x7_block", kwargs=dict(in_channels=3, out_channels=64, stride=2), x=(2, 3, 32, 32)),
    dict(name="conv3x3_stem_relu6", kind="conv3x3_block",
         kwargs=dict(in_channels=3, out_channels=32, stride=2, activation="relu6"), x=(2, 3, 32, 32)),
    dict(name="convblock_asym_pad_bias", kind="ConvBlock",
         kwargs=dict(in_channels=16, out_channels=32, kernel_size=3, stride=2, padding=(0, 1, 0, 1), bias=True),
         x=(2, 16, 12, 12)),
    dict(name="convblock_nonorm_bias", kind="ConvBlock",
         kwargs=dict(in_channels=16, out_channels=16, kernel_size=1, bias=True, normalization=None, activation=None),
         x=(2, 16, 6, 6)),
    dict(name="dwconv3x3_s1_relu6", kind="dwconv3x3_block",
         kwargs=dict(in_channels=144, out_channels=144, activation="relu6"), x=(2, 144, 14, 14)),
    dict(name="dwconv3x3_s2_relu6", kind="dwconv3x3_block",
         kwargs=dict(in_channels=96, out_channels=96, stride=2, activation="relu6"), x=(2, 96, 16, 16)),
    dict(name="dwconv3x3_7x7", kind="dwconv3x3_block",
         kwargs=dict(in_channels=960, out_channels=960, activation="relu6"), x=(2, 960, 7, 7)),
    dict(name="dwconv5x5_s1", kind="dwconv5x5_block", kwargs=dict(in_channels=40, out_channels=40), x=(2, 40, 10, 10)),
    dict(name="gconv3x3_g32_cg4", kind="conv3x3_block",
         kwargs=dict(in_channels=128, out_channels=128, groups=32), x=(2, 128, 8, 8)),
    dict(name="gconv3x3_g32_cg8_s2", kind="conv3x3_block",
         kwargs=dict(in_channels=256, out_channels=256, stride=2, groups=32), x=(1, 256, 8, 8)),
    dict(name="gconv3x3_g32_cg16", kind="conv3x3_block",
         kwargs=dict(in_channels=512, out_channels=512, groups=32), x=(1, 512, 7, 7)),
    dict(name="gconv3x3_g4_cg32", kind="conv3x3_block",
         kwargs=dict(in_channels=128, out_channels=128, groups=4), x=(2, 128, 7, 7)),
    dict(name="res_init_block", kind="ResInitBlock", kwargs=dict(in_channels=3, out_channels=64), x=(2, 3, 32, 32)),
    dict(name="res_unit_bottleneck_s2", kind="ResUnit",
         kwargs=dict(in_channels=64, out_channels=128, stride=2, bottleneck=True, conv1_stride=True), x=(2, 64, 14, 14)),
    dict(name="res_unit_bottleneck_id", kind="ResUnit",
         kwargs=dict(in_channels=128, out_channels=128, stride=1, bottleneck=True, conv1_stride=True), x=(2, 128, 7, 7)),
    dict(name="res_unit_basic_s2", kind="ResUnit",
         kwargs=dict(in_channels=32, out_channels=64, stride=2, bottleneck=False, conv1_stride=False), x=(2, 32, 14, 14)),
    dict(name="linear_bottleneck_res", kind="LinearBottleneck",
         kwargs=dict(in_channels=32, out_channels=32, stride=1, expansion=True, remove_exp_conv=False,
                     activation="relu6"), x=(2, 32, 14, 14)),
    dict(name="linear_bottleneck_s2", kind="LinearBottleneck",
         kwargs=dict(in_channels=24, out_channels=32, stride=2, expansion=True, remove_exp_conv=False,
                     activation="relu6"), x=(2, 24, 14, 14)),
    dict(name="resnext_unit_s2", kind="ResNeXtUnit",
         kwargs=dict(in_channels=128, out_channels=256, stride=2, cardinality=32, bottleneck_width=4), x=(2, 128, 14, 14)),
    dict(name="se_block", kind="SEBlock", kwargs=dict(channels=64), x=(2, 64, 7, 7)),
    dict(name="seres_unit", kind="SEResUnit",
         kwargs=dict(in_channels=64, out_channels=128, stride=2, bottleneck=True, conv1_stride=True), x=(2, 64, 14, 14)),
    # MobileNetV3 pieces (appended: the per-case seeds are index based)
    dict(name="dwconv5x5_s2_hswish", kind="dwconv5x5_block",
         kwargs=dict(in_channels=96, out_channels=96, stride=2, activation="hswish"), x=(2, 96, 15, 15)),
    dict(name="dwconv5x5_c72_relu", kind="dwconv5x5_block", kwargs=dict(in_channels=72, out_channels=72), x=(2, 72, 9, 9)),
    dict(name="conv1x1_c40_c120_hswish", kind="conv1x1_block",
         kwargs=dict(in_channels=40, out_channels=120, activation="hswish"), x=(2, 40, 14, 14)),
    dict(name="conv1x1_c48_c96_swish", kind="conv1x1_block",
         kwargs=dict(in_channels=48, out_channels=96, activation="swish"), x=(2, 48, 7, 7)),
    dict(name="se_block_r4_round_hsigmoid", kind="SEBlock",
         kwargs=dict(channels=72, reduction=4, round_mid=True, out_activation="hsigmoid"), x=(2, 72, 7, 7)),
    dict(name="mbv3_unit_k5_se_res", kind="MobileNetV3Unit",
         kwargs=dict(in_channels=40, out_channels=40, exp_channels=120, stride=1, use_kernel3=False, activation="hswish",
                     use_se=True), x=(2, 40, 14, 14)),
    dict(name="mbv3_unit_k3_s2_relu", kind="MobileNetV3Unit",
         kwargs=dict(in_channels=16, out_channels=24, exp_channels=64, stride=2, use_kernel3=True, activation="relu",
                     use_se=False), x=(2, 16, 16, 16)),
    dict(name="mbv3_unit_noexp_se_s2", kind="MobileNetV3Unit",
         kwargs=dict(in_channels=16, out_channels=16, exp_channels=16, stride=2, use_kernel3=True, activation="relu",
                     use_se=True), x=(2, 16, 16, 16)),
    # EfficientNet pieces; `bn_eps` / `activation` are turned into the lambda generators by build_block
    dict(name="effi_init_tf_even", kind="EffiInitBlock",
         kwargs=dict(in_channels=3, out_channels=32, bn_eps=1e-3, activation="swish", tf_mode=True), x=(2, 3, 32, 32)),
    dict(name="effi_init_tf_odd", kind="EffiInitBlock",
         kwargs=dict(in_channels=3, out_channels=32, bn_eps=1e-3, activation="swish", tf_mode=True), x=(2, 3, 33, 33)),
    dict(name="effi_dws_unit_tf", kind="EffiDwsConvUnit",
         kwargs=dict(in_channels=32, out_channels=16, stride=1, bn_eps=1e-3, activation="swish", tf_mode=True), x=(2, 32, 14, 14)),
    dict(name="effi_inv_res_k5_s2_tf_odd", kind="EffiInvResUnit",
         kwargs=dict(in_channels=24, out_channels=40, kernel_size=5, stride=2, exp_factor=6, se_factor=4, bn_eps=1e-3,
                     activation="swish", tf_mode=True), x=(2, 24, 15, 15)),
    dict(name="effi_inv_res_k3_s2_tf_nonsquare", kind="EffiInvResUnit",
         kwargs=dict(in_channels=16, out_channels=24, kernel_size=3, stride=2, exp_factor=6, se_factor=4, bn_eps=1e-3,
                     activation="swish", tf_mode=True), x=(2, 16, 14, 9)),
    dict(name="effi_inv_res_k5_res", kind="EffiInvResUnit",
         kwargs=dict(in_channels=40, out_channels=40, kernel_size=5, stride=1, exp_factor=6, se_factor=4, bn_eps=1e-5,
                     activation="swish", tf_mode=False), x=(2, 40, 10, 10)),
    # pre-activation blocks
    dict(name="pre_conv3x3_s2", kind="pre_conv3x3_block", kwargs=dict(in_channels=32, out_channels=64, stride=2), x=(2, 32, 14, 14)),
    dict(name="pre_conv1x1", kind="pre_conv1x1_block", kwargs=dict(in_channels=64, out_channels=32), x=(2, 64, 7, 7)),
    dict(name="preres_unit_bottleneck_s2", kind="PreResUnit",
         kwargs=dict(in_channels=64, out_channels=128, stride=2, bottleneck=True, conv1_stride=True), x=(2, 64, 14, 14)),
    dict(name="preres_unit_basic_id", kind="PreResUnit",
         kwargs=dict(in_channels=64, out_channels=64, stride=1, bottleneck=False, conv1_stride=False), x=(2, 64, 7, 7)),
    dict(name="preres_init_block", kind="PreResInitBlock", kwargs=dict(in_channels=3, out_channels=64), x=(2, 3, 32, 32)),
    dict(name="preres_activation", kind="PreResActivation", kwargs=dict(in_channels=128), x=(2, 128, 7, 7)),
    # DenseNet pieces
    dict(name="dense_unit", kind="DenseUnit", kwargs=dict(in_channels=64, out_channels=96, dropout_rate=0.0), x=(2, 64, 14, 14)),
    dict(name="dense_unit_c160", kind="DenseUnit", kwargs=dict(in_channels=160, out_channels=192, dropout_rate=0.0), x=(2, 160, 7, 7)),
    dict(name="dense_transition", kind="TransitionBlock", kwargs=dict(in_channels=256, out_channels=128), x=(2, 256, 14, 14)),
    # ShuffleNetV2 units: 116 = 2 x 58 channels (not multiples of 8), split / concat / shuffle
    dict(name="shuffle_unit", kind="ShuffleUnit",
         kwargs=dict(in_channels=116, out_channels=116, downsample=False, use_se=False, use_residual=False), x=(2, 116, 14, 14)),
    dict(name="shuffle_unit_down", kind="ShuffleUnit",
         kwargs=dict(in_channels=24, out_channels=116, downsample=True, use_se=False, use_residual=False), x=(2, 24, 28, 28)),
    dict(name="shuffle_init_block", kind="ShuffleInitBlock", kwargs=dict(in_channels=3, out_channels=24), x=(2, 3, 34, 34)),
]


# ---- branch / merge containers, interpolation, stand-alone BN + activation ------------------------------------
# name -> input shape. The blocks are built by `build_f4_block` from a namespace of constructors, so that the fixture generator
# (the reference's classes) and the tests (pytorchcv_amd's) assemble literally the same structure.
F4_CASES = {
    "norm_activation": (2, 64, 9, 9),
    "interp_bilinear_up2": (2, 32, 7, 9),
    "interp_bilinear_noalign_size": (1, 16, 6, 7),
    "interp_nearest_up2": (2, 24, 5, 6),
    "interp_nearest_ignores_up_false": (1, 16, 6, 5),          # reference tutti.py:232-238: F.interpolate(scale_factor=...) - `up` is not consulted
    "interp_nearest_ignores_out_size": (1, 8, 4, 7),           # ... nor `out_size`
    "interp_bilinear_down2": (1, 24, 12, 10),
    "concurrent_cat": (2, 32, 10, 10),
    "concurrent_cat_pool": (2, 32, 9, 11),
    "concurrent_sum": (2, 32, 10, 10),
    "seq_concurrent": (2, 16, 8, 8),
    "channel_shuffle_g2": (2, 48, 6, 5),
}


def build_f4_block(name, ns):
    """`ns`: object with Concurrent, SequentialConcurrent, NormActivation, InterpolationBlock, ChannelShuffle, conv1x1_block,
    conv3x3_block, Sequential, MaxPool (3x3 / stride 1 / pad 1 factory)."""
    if name == "norm_activation":
        return ns.NormActivation(in_channels=64)
    if name == "interp_bilinear_up2":
        return ns.InterpolationBlock(scale_factor=2)
    if name == "interp_bilinear_noalign_size":
        return ns.InterpolationBlock(scale_factor=None, out_size=(13, 10), align_corners=False)
    if name == "interp_nearest_up2":
        return ns.InterpolationBlock(scale_factor=2, mode="nearest", align_corners=None)
    if name == "interp_nearest_ignores_up_false":
        return ns.InterpolationBlock(scale_factor=2, mode="nearest", align_corners=None, up=False)
    if name == "interp_nearest_ignores_out_size":
        return ns.InterpolationBlock(scale_factor=3, out_size=(5, 5), mode="nearest", align_corners=None)
    if name == "interp_bilinear_down2":
        return ns.InterpolationBlock(scale_factor=2, up=False)
    if name == "concurrent_cat":
        blk = ns.Concurrent()
        blk.add_module("branch1", ns.conv1x1_block(in_channels=32, out_channels=16))
        blk.add_module("branch2", ns.conv3x3_block(in_channels=32, out_channels=24))
        b3 = ns.Sequential()
        b3.add_module("conv1", ns.conv1x1_block(in_channels=32, out_channels=8))
        b3.add_module("conv2", ns.conv3x3_block(in_channels=8, out_channels=32))
        blk.add_module("branch3", b3)
        return blk
    if name == "concurrent_cat_pool":
        blk = ns.Concurrent()
        blk.add_module("branch1", ns.conv3x3_block(in_channels=32, out_channels=16))
        b2 = ns.Sequential()
        b2.add_module("pool", ns.MaxPool())
        b2.add_module("conv", ns.conv1x1_block(in_channels=32, out_channels=8))
        blk.add_module("branch2", b2)
        blk.add_module("branch3", ns.MaxPool())
        return blk
    if name == "concurrent_sum":
        blk = ns.Concurrent(merge_type="sum")
        blk.add_module("branch1", ns.conv1x1_block(in_channels=32, out_channels=32))
        blk.add_module("branch2", ns.conv3x3_block(in_channels=32, out_channels=32))
        blk.add_module("branch3", ns.MaxPool())
        return blk
    if name == "seq_concurrent":
        blk = ns.SequentialConcurrent()
        blk.add_module("conv1", ns.conv3x3_block(in_channels=16, out_channels=16))
        blk.add_module("conv2", ns.conv3x3_block(in_channels=16, out_channels=16))
        return blk
    if name == "channel_shuffle_g2":
        return ns.ChannelShuffle(channels=48, groups=2)
    raise KeyError(name)
