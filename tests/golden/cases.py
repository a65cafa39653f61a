"""
    Block-level parity cases shared by the fixture generator (tests/golden/make_golden.py, which builds them
    from the imported reference) and by the tests (which build the same blocks from pytorchcv_amd / the oracle).
    Each case: name, kind (constructor in the reference), kwargs, input shape. Data only.
"""

MODELS = ["resnet18", "resnet50", "mobilenetv2_w1", "resnext101_32x4d", "seresnet50", "seresnext50_32x4d", "mobilenet_w1"]

# known answers from the reference's own asserts (resnet.py:983,988; mobilenetv2.py:436; resnext.py:547;
# model_metainfos.csv:81) and SURVEY.md section 8a16 (state_dict key counts)
PARAM_COUNTS = {"resnet18": 11689512, "resnet50": 25557032, "mobilenetv2_w1": 3504960,
                "resnext101_32x4d": 44177704, "seresnet50": 28088024, "seresnext50_32x4d": 27559896,
                "mobilenet_w1": 4231976}
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
]
