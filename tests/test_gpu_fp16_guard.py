"""
GPU tests of the fp16 range guard (include/pcv_amd.h: pcv_fp16_guard_begin / _end / pcv_fp16_overflow_count).

fp16 is the default 16-bit storage type of the depthwise-separable families (engine.compute_dtype_of) because bf16 misses the
north-star 1e-2 there; what fp16 gives up is exponent range. The reference computes in fp32 (e.g. `MobileNetV2.forward`,
pytorchcv/models/mobilenetv2.py:152-156), where these magnitudes are unremarkable, so a value that leaves fp16's range must
never turn into a plausible result: every kernel that rounds to fp16 counts such values, and a guarded forward whose count
moved returns NaN. Each kernel family is driven over the edge here by blowing up the BatchNorm gain of its last convolution.
"""

import pytest
import torch
import torch.nn as nn
import util

pytestmark = pytest.mark.gpu

BIG = 3.0e6            # BN gain factor: outputs of O(1) become O(1e6) >> 65504


def _blow_up(block, key_suffix="bn.weight", which=-1):
    """Multiply one BatchNorm gain (by default the last one in module order) by BIG; returns the key it touched."""
    keys = [k for k in block.state_dict() if k.endswith(key_suffix)]
    k = keys[which]
    with torch.no_grad():
        block.state_dict()[k].mul_(BIG)
    return k


def _count(dev):
    from pytorchcv_amd import engine
    return engine.fp16_overflow_count(dev)


def _check_block(blk, x, dev, tuning=None, expect_overflow=True):
    """fp16: the guarded block returns NaN everywhere and the counter moves; bf16 on the same weights stays finite."""
    import pytorchcv_amd
    blk = blk.eval().to(dev)
    x = x.to(dev)
    with torch.no_grad():
        pytorchcv_amd.set_compute_dtype(blk, "bf16")
        y_bf = blk(x)
        torch.cuda.synchronize()
        before = _count(dev)
        pytorchcv_amd.set_compute_dtype(blk, "fp16")
        if tuning:
            with util.tuning(**tuning):
                y = blk(x)
                torch.cuda.synchronize()
        else:
            y = blk(x)
    torch.cuda.synchronize()
    after = _count(dev)
    assert bool(torch.isfinite(y_bf).all()), "the bf16 run of the same weights must be finite (fp32's exponent range)"
    if expect_overflow:
        assert after > before, "no fp16 overflow was counted"
        assert bool(torch.isnan(y).all()), "the guarded fp16 result must be NaN everywhere"
        assert float(y_bf.abs().max()) > 65504.0
    else:
        assert after == before
        assert bool(torch.isfinite(y).all())


def _synth(blk, seed=7):
    blk.load_state_dict(util.synth_state_dict(blk.state_dict(), seed=seed))
    return blk


def test_guard_is_quiet_inside_range(cuda_device):
    from pytorchcv_amd.models.common.conv import conv1x1_block
    blk = _synth(conv1x1_block(in_channels=32, out_channels=48, activation=None))
    _check_block(blk, util.synth_input(2, 32, 9, 9, seed=1), cuda_device, expect_overflow=False)


def test_generic_conv_kernel(cuda_device):
    """igemm_conv.hpp epilogue (1x1, no activation)."""
    from pytorchcv_amd.models.common.conv import conv1x1_block
    blk = _synth(conv1x1_block(in_channels=32, out_channels=48, activation=None))
    _blow_up(blk)
    _check_block(blk, util.synth_input(2, 32, 9, 9, seed=1), cuda_device)


def test_relu6_bounded_output_is_not_an_overflow(cuda_device):
    """ReLU6 clamps in fp32 BEFORE the rounding: the stored value is 6, as in the reference - nothing to report."""
    from pytorchcv_amd.models.common.conv import conv1x1_block
    from pytorchcv_amd.models.common.activ import lambda_relu6
    blk = _synth(conv1x1_block(in_channels=32, out_channels=48, activation=lambda_relu6()))
    _blow_up(blk)
    import pytorchcv_amd
    blk = pytorchcv_amd.set_compute_dtype(blk.eval().to(cuda_device), "fp16")
    before = _count(cuda_device)
    with torch.no_grad():
        y = blk(util.synth_input(2, 32, 9, 9, seed=1).to(cuda_device))
    torch.cuda.synchronize()
    assert _count(cuda_device) == before and bool(torch.isfinite(y).all()) and float(y.max()) == 6.0


@pytest.mark.parametrize("mode", ["d3x3", "d1x1"])
def test_eight_wave_kernel(mode, cuda_device):
    """d3q_conv.hpp epilogue: dense 3x3 mode and 1x1 mode (forced tile shape so that the small fixture takes the kernel)."""
    from pytorchcv_amd.models.common.conv import conv1x1_block, conv3x3_block
    if mode == "d3x3":
        blk, x = _synth(conv3x3_block(in_channels=64, out_channels=64)), util.synth_input(2, 64, 28, 28, seed=2)
    else:
        blk, x = _synth(conv1x1_block(in_channels=256, out_channels=128, activation=None)), util.synth_input(2, 256, 14, 14, seed=2)
    _blow_up(blk)
    _check_block(blk, x, cuda_device, tuning={mode: 1})


@pytest.mark.parametrize("pool", [False, True])
def test_stem_kernel(pool, cuda_device):
    """stem_conv.hpp: plain epilogue, and the packed-pool epilogue of ResInitBlock (7x7/2 + MaxPool)."""
    from pytorchcv_amd.models.common.conv import conv7x7_block
    from pytorchcv_amd.models.resnet import ResInitBlock
    blk = _synth(ResInitBlock(in_channels=3, out_channels=64) if pool else conv7x7_block(in_channels=3, out_channels=64, stride=2))
    _blow_up(blk)
    _check_block(blk, util.synth_input(2, 3, 64, 64, seed=3), cuda_device)


def test_input_conversion(cuda_device):
    """An image value beyond fp16's range is caught where it is rounded (pcv_nchw_to_nhwc / the NCHW stem's patch conversion)."""
    from pytorchcv_amd.models.common.conv import conv3x3_block
    blk = _synth(conv3x3_block(in_channels=3, out_channels=32, stride=2))
    x = util.synth_input(2, 3, 32, 32, seed=3)
    x[1, 2, 5, 7] = 1.0e5
    import pytorchcv_amd
    blk = pytorchcv_amd.set_compute_dtype(blk.eval().to(cuda_device), "fp16")
    before = _count(cuda_device)
    with torch.no_grad():
        y = blk(x.to(cuda_device))
    torch.cuda.synchronize()
    assert _count(cuda_device) > before and bool(torch.isnan(y).all())


@pytest.mark.parametrize("k,stride", [(3, 1), (3, 2), (5, 1), (5, 2)])
def test_depthwise_kernels(k, stride, cuda_device):
    """dwconv.hpp: 3x3 register-window kernel and 5x5 row-streaming kernel."""
    from pytorchcv_amd.models.common.conv import dwconv3x3_block, dwconv5x5_block
    ctor = dwconv3x3_block if k == 3 else dwconv5x5_block
    blk = _synth(ctor(in_channels=40, out_channels=40, stride=stride, activation=None))
    _blow_up(blk)
    _check_block(blk, util.synth_input(2, 40, 19, 17, seed=4), cuda_device)


@pytest.mark.parametrize("stride,width", [(1, 128), (2, 256), (1, 1024)])
def test_grouped_kernels(stride, width, cuda_device):
    """gconv3x3.hpp (stride 1, 4 channels per group) and gconv3x3r.hpp (stride 2; 32 channels per group)."""
    from pytorchcv_amd.models.common.conv import conv3x3_block
    blk = _synth(conv3x3_block(in_channels=width, out_channels=width, stride=stride, groups=32))
    _blow_up(blk)
    _check_block(blk, util.synth_input(2, width, 14, 14, seed=5), cuda_device)


@pytest.mark.parametrize("cm,which", [(64, "first"), (64, "second"), (128, "first"), (256, "second")])
def test_fused_pair_kernels(cm, which, cuda_device):
    """pair1x1.hpp / wpair1x1.hpp: epilogue 1 (y1, written AND fed to the second GEMM) and epilogue 2 (y2)."""
    import pytorchcv_amd
    from pytorchcv_amd import engine
    from pytorchcv_amd.models.common.conv import conv1x1_block, conv_block_pair
    first = _synth(conv1x1_block(in_channels=cm, out_channels=4 * cm, activation=None), 21)
    second = _synth(conv1x1_block(in_channels=4 * cm, out_channels=cm), 22)
    _blow_up(first if which == "first" else second)
    first, second = [pytorchcv_amd.set_compute_dtype(b.eval().to(cuda_device), "fp16") for b in (first, second)]
    g = torch.Generator().manual_seed(5)
    N, H, W = 3, 13, 11
    x = engine.NHWC(torch.randn((N, H, W, cm), generator=g).to(cuda_device).half(), N, H, W, cm)
    r = engine.NHWC(torch.randn((N, H, W, 4 * cm), generator=g).to(cuda_device).half(), N, H, W, 4 * cm)
    before = _count(cuda_device)
    with torch.no_grad():
        pair = conv_block_pair(first, x, r, nn.ReLU(), second)
    assert pair is not None
    torch.cuda.synchronize()
    assert _count(cuda_device) > before
    y = pair[0 if which == "first" else 1].t.float()
    assert bool(torch.isinf(y).any())               # (unguarded handles: the raw fp16 result shows the infinities the counter reports)


@pytest.mark.parametrize("kernel", ["wave", "block"])
@pytest.mark.parametrize("shape", [(2, 28, 28, 24, 24, 1), (2, 28, 28, 32, 64, 2), (2, 14, 14, 96, 96, 1)],
                         ids=["24-144-24", "32-192-64s2", "96-576-96"])
def test_fused_inverted_residual_unit(shape, kernel, cuda_device):
    """mbw.hpp / mbconv.hpp: the projection (no activation) leaves fp16's range; the ReLU6-bounded expand / depthwise stages
    cannot. Behind the unit a ReLU6 convolution would clamp the infinity to 6: exactly the case the guard exists for."""
    from pytorchcv_amd.models.mobilenetv2 import LinearBottleneck
    from pytorchcv_amd.models.common.activ import lambda_relu6
    N, H, W, cin, cout, stride = shape
    if kernel == "block" and (cout > 32 or W < 24):
        pytest.skip("the block-tile kernel does not take this shape")
    unit = _synth(LinearBottleneck(in_channels=cin, out_channels=cout, stride=stride, expansion=True, remove_exp_conv=False,
                                   activation=lambda_relu6()), 9)
    assert _blow_up(unit).startswith("conv3.")
    _check_block(unit, util.synth_input(N, cin, H, W, seed=6), cuda_device, tuning={"mbw": 1 if kernel == "wave" else 0})


def test_elementwise_kernels(cuda_device):
    """pcv_bn_act (pre-activation) and pcv_se_scale (here through engine.add: 60000 + 60000)."""
    from pytorchcv_amd import engine
    from pytorchcv_amd.models.preresnet import PreResActivation
    blk = _synth(PreResActivation(in_channels=32))
    _blow_up(blk)
    _check_block(blk, util.synth_input(2, 32, 9, 9, seed=1), cuda_device)
    a = engine.NHWC(torch.full((1, 4, 4, 16), 60000.0, dtype=torch.float16, device=cuda_device), 1, 4, 4, 16)
    before = _count(cuda_device)
    y = engine.add(a, a)
    torch.cuda.synchronize()
    assert _count(cuda_device) > before and bool(torch.isinf(y.t.float()).all())


@pytest.mark.parametrize("name,key", [
    ("mobilenetv2_w1", "features.stage3.unit1.conv3.bn.weight"),       # stride-2 unit: no skip; the next expand + ReLU6 hides it
    ("mobilenetv2_w1", "features.stage4.unit6.conv3.bn.weight"),       # a wide fused unit (96 projected channels)
    ("mobilenetv3_small_w1", "features.stage2.unit1.conv2.bn.weight"),       # conv2 = the projection there
    ("efficientnet_b0", "features.stage3.unit1.conv3.bn.weight"),
    ("resnet50", "features.stage1.unit2.body.conv2.bn.weight"),
])
def test_whole_net_overflow_gives_nan_logits_not_plausible_ones(name, key, cuda_device):
    """A net whose activations leave fp16's range somewhere in the middle: without the guard MobileNetV2 returns FINITE logits
    (the next unit's ReLU6 clamps the infinity to 6) - with it, NaN. The same weights in bf16 give finite logits. Eager and
    through a 2-lane hipGraph."""
    import pytorchcv_amd
    from pytorchcv_amd.model_provider import get_model
    from pytorchcv_amd.graph import capture
    net = get_model(name).eval()
    sd = util.model_state(name, net.state_dict())
    sd[key] = sd[key] * BIG
    net.load_state_dict(sd, strict=True)
    net = net.to(cuda_device)
    x = util.synth_input(4, seed=11).to(cuda_device)
    with torch.no_grad():
        y_bf = pytorchcv_amd.set_compute_dtype(net, "bf16")(x).clone()
        torch.cuda.synchronize()
        before = _count(cuda_device)
        pytorchcv_amd.set_compute_dtype(net, "fp16")
        y = net(x).clone()
        torch.cuda.synchronize()
        mid = _count(cuda_device)
        g = capture(net, x, lanes=2)
        y_g = g(x, clone=True)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(y_bf).all())
    assert mid > before and _count(cuda_device) > mid
    assert bool(torch.isnan(y).all()) and bool(torch.isnan(y_g).all())
    # a healthy forward on the same context afterwards is not poisoned (the counter is compared, never reset)
    ok = get_model("mobilenetv2_w1").eval()
    ok.load_state_dict(util.model_state("mobilenetv2_w1", ok.state_dict()), strict=True)
    with torch.no_grad():
        y_ok = ok.to(cuda_device)(x)
    assert bool(torch.isfinite(y_ok).all())


def test_guard_api_argument_checks(cuda_device):
    import ctypes
    from pytorchcv_amd import _lib
    L, ctx = _lib.lib(), _lib.ctx_for(0)
    assert L.pcv_fp16_guard_begin(ctx, None, None) == -1
    assert b"slot" in L.pcv_last_error(ctx)
    slot = torch.zeros(1, dtype=torch.int32, device=cuda_device)
    y = torch.zeros(8, dtype=torch.float32, device=cuda_device)
    assert L.pcv_fp16_guard_end(ctx, ctypes.c_void_p(slot.data_ptr()), ctypes.c_void_p(y.data_ptr()), 0, None) == -1
    assert L.pcv_fp16_overflow_count(ctx, None, None) == -1
    # begin / end with nothing in between leave the result alone
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert L.pcv_fp16_guard_begin(ctx, ctypes.c_void_p(slot.data_ptr()), st) == 0
    assert L.pcv_fp16_guard_end(ctx, ctypes.c_void_p(slot.data_ptr()), ctypes.c_void_p(y.data_ptr()), 8, st) == 0
    torch.cuda.synchronize()
    assert bool((y == 0).all())


@pytest.mark.gpu
def test_submodule_called_with_a_tensor_runs_in_the_family_type_and_is_guarded(cuda_device):
    """`net.features(x)` (any part of an fp16-family net called with an NCHW tensor) resolves "auto" like `net(x)` - the
    result equals the explicit-fp16 handle path bit for bit - and the pre-activation blocks' tensor entry carries the range guard."""
    import pytorchcv_amd
    from pytorchcv_amd import engine
    from pytorchcv_amd.model_provider import get_model
    from pytorchcv_amd.models.common.conv import pre_conv3x3_block
    net = get_model("mobilenetv2_w1").eval()
    net.load_state_dict(util.model_state("mobilenetv2_w1", net.state_dict()), strict=True)
    net = net.to(cuda_device)
    x = util.synth_input(2, seed=5).to(cuda_device)
    unit = net.features.stage2.unit1
    with torch.no_grad():
        a = engine.from_nchw(x, "fp16", stem=True)
        h = net.features.init_block(a)
        want = engine.to_nchw(unit(net.features.stage1(h)))
        xin = engine.to_nchw(net.features.stage1(h))            # fp32 NCHW copy of fp16 values: converts back exactly
        got = unit(xin)
    assert engine.compute_dtype_of(unit) == "fp16"
    assert torch.equal(got, want)
    blk = pre_conv3x3_block(in_channels=16, out_channels=16, return_preact=True)
    sd = util.synth_state_dict(blk.state_dict(), seed=9)
    sd["conv.weight"] = sd["conv.weight"] * BIG
    blk.load_state_dict(sd)
    blk = pytorchcv_amd.set_compute_dtype(blk.eval().to(cuda_device), "fp16")
    with torch.no_grad():
        y, pre = blk(util.synth_input(2, 16, 12, 12, seed=3).to(cuda_device))
    torch.cuda.synchronize()
    assert bool(torch.isnan(y).all()), "an fp16 overflow inside a pre-activation block called with a tensor must poison its result"
