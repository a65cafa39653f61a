"""
GPU parity, whole nets: `get_model(name)` + the fixture weights, forward through the C ABI on MI355X, against the
golden logits of the imported reference and the oracle.

  fp32       : |logits - golden| <= 1e-3, top-1 identical                     (north-star fp32 bound)
  16 bit     : the north-star bound - max |logits - fp32 golden of the imported reference| <= 1e-2, top-1 identical - is asserted
               as a literal 1e-2 for EVERY fixture net in the mode it runs in by default ("auto": bf16, or fp16 for the
               depthwise-separable families - engine.compute_dtype_of) and for fp16 on every net;
  bf16, fp16 : |logits - quantisation-matched oracle| <= 1e-2, top-1 identical to the oracle's and the reference's, on every
               net in both types (kernel correctness; a family's non-default 16-bit type is held to this bound only:
               DESIGN.md section 3 has the attribution of bf16's 1.3e-2 .. 2.2e-2 on the depthwise nets).
"""

import pytest
import torch
import util
from oracle import refnet

pytestmark = pytest.mark.gpu

NORTH_STAR_16BIT = 1e-2          # BASELINE.json north_star: "within ... 1e-2 (bf16)" of the reference forward
_FP16_FAMILIES = ("mobilenetv2", "mobilenetv3", "efficientnet")


def _auto_dtype(name):
    return "fp16" if name.startswith(_FP16_FAMILIES) else "bf16"


def _net(name, dtype, dev):
    import pytorchcv_amd
    from pytorchcv_amd.model_provider import get_model
    net = get_model(name).eval()
    net.load_state_dict(util.model_state(name, net.state_dict()), strict=True)
    net = net.to(dev)
    return net if dtype is None else pytorchcv_amd.set_compute_dtype(net, dtype)


@pytest.mark.parametrize("name", util.MODELS)
def test_model_default_mode_within_north_star_bound(name, cuda_device, monkeypatch):
    """`get_model(name)` as a user gets it - no dtype chosen anywhere - against the fp32 golden of the imported reference
    (reference forward: e.g. pytorchcv/models/mobilenetv2.py:152-156, resnet.py:333-337): max |d| <= 1e-2, top-1 identical,
    no fp16 overflow counted."""
    from pytorchcv_amd import engine
    monkeypatch.delenv("PCV_AMD_DTYPE", raising=False)
    logits, ids = util.model_golden(name)
    net = _net(name, None, cuda_device)
    assert engine.compute_dtype_of(net) == _auto_dtype(name)
    before = engine.fp16_overflow_count(cuda_device)
    with torch.no_grad():
        y = net(util.images(ids).to(cuda_device))
    torch.cuda.synchronize()
    assert engine.fp16_overflow_count(cuda_device) == before
    y = y.cpu()
    raw = float((y - logits).abs().max())
    print("{} default ({}): vs fp32 golden {:.3e}".format(name, _auto_dtype(name), raw))
    assert bool(torch.isfinite(y).all())
    assert torch.equal(y.argmax(1), logits.argmax(1)), "top-1 differs from the reference forward"
    assert raw <= NORTH_STAR_16BIT, "{}: max |d| vs the fp32 reference golden {:.3e} > 1e-2".format(name, raw)


@pytest.mark.parametrize("name", util.MODELS)
def test_model_fp32_matches_reference_golden(name, cuda_device):
    logits, ids = util.model_golden(name)
    net = _net(name, "fp32", cuda_device)
    with torch.no_grad():
        y = net(util.images(ids).to(cuda_device))
    torch.cuda.synchronize()
    y = y.cpu()
    assert y.shape == (len(ids), 1000) and y.dtype == torch.float32
    err = float((y - logits).abs().max())
    print("{} fp32: max|d|={:.3e}".format(name, err))
    assert err <= 1e-3
    assert torch.equal(y.argmax(1), logits.argmax(1))


_BF16_KNOWN_MISS_BOUND = 3e-2      # bf16 logits of the fp16-default families against the reference's fp32 forward (measured 1.3e-2 .. 2.2e-2)


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("name", util.MODELS)
def test_model_16bit_matches_quantisation_matched_oracle(name, dtype, cuda_device):
    logits, ids = util.model_golden(name)
    net = _net(name, dtype, cuda_device)
    x = util.images(ids)
    with torch.no_grad():
        y = net(x.to(cuda_device))
    torch.cuda.synchronize()
    y = y.cpu()
    sd = {k: v.cpu() for k, v in net.state_dict().items()}
    ref = refnet.forward(name, sd, x, quant=dtype)
    err = float((y - ref).abs().max())
    raw = float((y - logits).abs().max())
    print("{} {}: vs matched oracle {:.3e}; vs fp32 golden {:.3e}; top-1 vs golden equal: {}".format(
        name, dtype, err, raw, bool(torch.equal(y.argmax(1), logits.argmax(1)))))
    assert err <= 1e-2
    assert torch.equal(y.argmax(1), ref.argmax(1))
    assert torch.equal(y.argmax(1), logits.argmax(1)), "top-1 differs from the reference forward"
    if dtype == "fp16" or dtype == _auto_dtype(name):
        assert raw <= NORTH_STAR_16BIT, "{} {}: max |d| vs the fp32 reference golden {:.3e} > 1e-2".format(name, dtype, raw)
    else:
        # bf16 on a family whose default mode is fp16 (MobileNetV2 / V3, EfficientNet): the north star's literal 1e-2 is NOT met in
        # bf16 there (measured 1.3e-2 .. 2.2e-2: 8-bit-mantissa weights on [0, 6]-bounded activations) - which
        # is why "auto" is fp16 for them. Asserted as what it is, a known miss with a measured bound: a regression of the bf16 mode
        # on these families (> 3e-2) fails, and so does a silent improvement below 1e-2 (then bf16 may become their default again).
        assert NORTH_STAR_16BIT < raw <= _BF16_KNOWN_MISS_BOUND, \
            "{} bf16 vs the fp32 reference golden: {:.3e}, expected inside (1e-2, {:.0e}]".format(name, raw, _BF16_KNOWN_MISS_BOUND)


# BASELINE.json configs 2-4 (+ resnet18 at the same batch): the batches bench.py times. At these sizes every persistent kernel
# walks several rounds of tiles per block (ResNet-50's 3x3 layers: 1568 tiles on 512 slots) and the graph runs two batch lanes.
_HEADLINE = [("resnet50", 256), ("mobilenetv2_w1", 512), ("resnext101_32x4d", 256), ("resnet18", 256)]


@pytest.mark.parametrize("name,batch", _HEADLINE, ids=["{}-{}".format(n, b) for n, b in _HEADLINE])
def test_headline_batch_matches_fixture(name, batch, cuda_device):
    """The 4 golden images tiled to the full benchmark batch: every row of the eager forward AND of the 2-lane hipGraph
    replay is bit-identical to the 4-image forward (whose distance to the oracle / the reference golden the tests above
    bound), i.e. the multi-round tile schedules and the lane split compute exactly what the fixtures check.
    Reference behaviour: net(x) at any batch, pytorchcv/models/resnet.py:333-337."""
    from pytorchcv_amd.graph import capture
    logits, ids = util.model_golden(name)
    net = _net(name, "auto", cuda_device)               # the mode bench.py times: bf16, MobileNetV2 fp16
    x4 = util.images(ids).to(cuda_device)
    with torch.no_grad():
        y4 = net(x4).clone()
        x = x4.repeat(batch // 4, 1, 1, 1).contiguous()
        y_eager = net(x).clone()
        g = capture(net, x, lanes=2)
        assert g.lanes == 2
        y_graph = g(x, clone=True)
    torch.cuda.synchronize()
    want = y4.repeat(batch // 4, 1)
    assert y_eager.shape == (batch, 1000)
    assert torch.equal(y_eager, want), "eager full batch differs from the 4-image forward in {} rows".format(
        int((y_eager != want).any(1).sum()))
    assert torch.equal(y_graph, want), "2-lane graph differs from the 4-image forward in {} rows".format(
        int((y_graph != want).any(1).sum()))
    assert torch.equal(y_graph.argmax(1).cpu(), logits.argmax(1).repeat(batch // 4))      # top-1 of the reference forward
    del g, x, y_eager, y_graph
    torch.cuda.empty_cache()


def test_batch_independence_and_determinism(cuda_device):
    """Image i's logits do not depend on the rest of the batch, and a rerun is bit-identical."""
    net = _net("resnet18", "bf16", cuda_device)
    _, ids = util.model_golden("resnet18")
    x = util.images(ids).to(cuda_device)
    with torch.no_grad():
        y = net(x)
        y2 = net(x)
        y_single = torch.cat([net(x[i:i + 1]) for i in range(x.shape[0])])
    assert torch.equal(y, y2)
    assert torch.equal(y, y_single)


def test_odd_num_classes_and_in_channels(cuda_device):
    """num_classes not a multiple of 8 (ragged epilogue) and a 1-channel input (padded stem)."""
    import pytorchcv_amd
    from pytorchcv_amd.model_provider import get_model
    net = get_model("resnet18", num_classes=10, in_channels=1).eval()
    sd = util.synth_state_dict(net.state_dict(), seed=99)
    net.load_state_dict(sd)
    net = pytorchcv_amd.set_compute_dtype(net.to(cuda_device), "fp32")
    x = util.synth_input(2, 1, 224, 224, seed=3)
    with torch.no_grad():
        y = net(x.to(cuda_device)).cpu()
    # oracle with the same state dict (architecture differs from the named model only in the stem/classifier shapes)
    ref = refnet.resnet_forward(sd, x, blocks=18)
    assert y.shape == (2, 10)
    assert float((y - ref).abs().max()) <= 1e-3 * max(1.0, float(ref.abs().max()))


def test_batch_split_beyond_2gib_window(cuda_device):
    """One launch addresses its input through a 32-bit buffer window (< 2 GiB); larger batches are split by the host
    (PCV_ERR_TOO_LARGE -> ConvRunner._launch_range). The split result equals the per-chunk results bit for bit."""
    import pytorchcv_amd
    from pytorchcv_amd import engine
    from pytorchcv_amd.models.common.conv import conv1x1_block, dwconv3x3_block
    N, C, H = 36000, 32, 32                      # 36000*32*32*32*2 B = 2.36 GB > 2 GiB
    x = torch.empty((N, H, H, C), dtype=torch.bfloat16, device=cuda_device)
    base = torch.randn((500, H, H, C), device=cuda_device).to(torch.bfloat16)
    for i in range(0, N, 500):
        x[i:i + 500] = base
    xh = engine.NHWC(x, N, H, H, C)
    for blk in (conv1x1_block(in_channels=C, out_channels=16), dwconv3x3_block(in_channels=C, out_channels=C)):
        blk = blk.eval()
        blk.load_state_dict(util.synth_state_dict(blk.state_dict(), seed=3))
        blk = pytorchcv_amd.set_compute_dtype(blk.to(cuda_device), "bf16")
        with torch.no_grad():
            y = blk(xh).t
            y_small = blk(engine.NHWC(base, 500, H, H, C)).t
        assert y.shape[0] == N
        assert torch.equal(y[:500], y_small) and torch.equal(y[N - 500:], y_small) and torch.equal(y[17500:18000], y_small)
    del x, y
    torch.cuda.empty_cache()


def test_nchw_stem_beyond_2gib_window_falls_back(cuda_device):
    """An fp32 NCHW image batch of 2 GiB or more cannot be addressed by one launch of the NCHW stem: the lazy input handle converts
    itself (pcv_nchw_to_nhwc) and the stem runs on the converted tensor, split by the host where its own tensors exceed the window.
    Every image equals the small-batch result bit for bit."""
    import pytorchcv_amd
    from pytorchcv_amd import engine
    from pytorchcv_amd.models.common.conv import conv3x3_block
    N, H = 45000, 64                                  # 45000*3*64*64*4 B = 2.21 GB
    blk = conv3x3_block(in_channels=3, out_channels=32, stride=2).eval()
    blk.load_state_dict(util.synth_state_dict(blk.state_dict(), seed=3))
    blk = pytorchcv_amd.set_compute_dtype(blk.to(cuda_device), "bf16")
    base = util.synth_input(500, 3, H, H, seed=21).to(cuda_device)
    x = base.repeat(N // 500, 1, 1, 1).contiguous()
    assert x.numel() * 4 >= 2 ** 31
    with torch.no_grad():
        a = engine.network_input(x, "bf16")
        assert isinstance(a, engine.LazyNCHW)
        y = blk(a).t
        assert a.materialized, "the 2 GiB image batch must have gone through the layout kernel"
        small = engine.network_input(base, "bf16")
        y_small = blk(small).t
        assert not small.materialized
    assert y.shape[0] == N
    assert torch.equal(y[:500], y_small) and torch.equal(y[N - 500:], y_small) and torch.equal(y[22000:22500], y_small)
    del x, y, a
    torch.cuda.empty_cache()


def test_graph_replay_equals_eager(cuda_device):
    """hipGraph capture of the whole forward: bit-identical to eager, for new input contents too."""
    from pytorchcv_amd.graph import capture
    net = _net("resnet18", "bf16", cuda_device)
    _, ids = util.model_golden("resnet18")
    x = util.images(ids).to(cuda_device)
    with torch.no_grad():
        y_eager = net(x).clone()
        g = capture(net, x)
        y_graph = g(x, clone=True)
        x2 = torch.flip(x, dims=[0])
        y2_graph = g(x2, clone=True)
        y2_eager = net(x2)
    assert torch.equal(y_eager, y_graph)
    assert torch.equal(y2_eager, y2_graph)
    assert torch.equal(y2_graph, torch.flip(y_graph, dims=[0]))
    with pytest.raises(RuntimeError, match="captured for input shape"):
        g(x[:2])


@pytest.mark.parametrize("model,lanes", [(m, 3 if m == "mobilenetv2_w1" else 2) for m in util.MODELS])
def test_graph_batch_lanes_equal_single_lane(model, lanes, cuda_device):
    """The batch cut into independent graph branches (the kernels of one slice fill the tile-schedule tails of the other's):
    same logits, bit for bit, as the one-branch graph and as eager - also when the batch does not divide evenly."""
    from pytorchcv_amd.graph import capture
    net = _net(model, "auto", cuda_device)
    x = util.synth_input(7, seed=21).to(cuda_device)
    with torch.no_grad():
        y_eager = net(x).clone()
        g = capture(net, x, lanes=lanes)
        assert g.lanes == lanes
        y_lanes = g(x, clone=True)
        y_again = g(torch.flip(x, dims=[0]), clone=True)
    assert y_lanes.shape == y_eager.shape
    assert torch.equal(y_lanes, y_eager)
    assert torch.equal(y_again, torch.flip(y_eager, dims=[0]))


@pytest.mark.parametrize("model", ["resnet18", "mobilenetv2_w1", "seresnet50", "efficientnet_b0"])
def test_steps_in_flight_equal_eager(model, cuda_device):
    """PipelinedNet: consecutive batches in flight on alternating streams (two or three captured forwards with their own static
    buffers). DIFFERENT inputs on consecutive steps - so a slot that read another slot's buffers, or a step that started before its
    input had landed, shows - and every step's logits equal the eager forward of ITS input bit for bit; the `then` callback runs on
    the slot's stream behind the replay; a slot's output stays valid until that slot runs again."""
    from pytorchcv_amd.graph import PipelinedNet
    net = _net(model, "auto", cuda_device)
    xs = [util.synth_input(6, seed=40 + i).to(cuda_device) for i in range(7)]
    with torch.no_grad():
        want = [net(x).clone() for x in xs]
        for depth, lanes in ((2, 1), (3, 2)):
            p = PipelinedNet(net, xs[0], depth=depth, lanes=lanes)
            assert (p.depth, p.lanes) == (depth, lanes) and len(p.static_inputs) == depth
            got = [p(x, then=lambda y: y.clone()) for x in xs]       # the clone rides on the slot's stream
            p.synchronize()
            for i, (g, w) in enumerate(zip(got, want)):
                assert torch.equal(g, w), "step {} (slot {}) of depth {} differs from eager".format(i, i % depth, depth)
            # x = None: the slot replays what its static input holds - the caller fills the buffers in place
            for k, buf in enumerate(p.static_inputs):
                buf.copy_(xs[k + 1])
            torch.cuda.synchronize()
            first = p.next_slot                                      # the slots run in turn, continuing where the loop above stopped
            ys = [p(None) for _ in range(depth)]
            p.synchronize()
            for j in range(depth):
                assert torch.equal(ys[j], want[(first + j) % depth + 1])


def test_capture_best_picks_a_launcher_and_keeps_the_arithmetic(cuda_device):
    """capture_best: below batch 64 one graph with one lane; from 64 up whichever of {one graph x two lanes, two graphs in flight}
    replays faster on this device - either way the logits of a replay are the eager logits."""
    from pytorchcv_amd.graph import capture_best, GraphedNet, PipelinedNet
    net = _net("resnet18", "auto", cuda_device)
    with torch.no_grad():
        small = util.synth_input(8, seed=3).to(cuda_device)
        g = capture_best(net, small)
        assert isinstance(g, GraphedNet) and g.lanes == 1
        assert torch.equal(g(small, clone=True), net(small))
        big = util.synth_input(8, seed=4).to(cuda_device).repeat(8, 1, 1, 1).contiguous()
        b = capture_best(net, big, steps=6)
        assert isinstance(b, (GraphedNet, PipelinedNet))
        y = b(None)
        torch.cuda.synchronize()
        assert torch.equal(y, net(big))
