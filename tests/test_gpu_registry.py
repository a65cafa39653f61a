"""GPU smoke + parity over the wider registry: variants that no golden fixture covers (other depths, widths, input sizes,
TF-same padding at odd sizes) run through the C ABI with seeded random weights and are compared with the oracle in fp32."""

import pytest
import torch
import util
from oracle import refnet

pytestmark = pytest.mark.gpu

# (name, input size, compare with the oracle?)   - the oracle forward of the largest nets takes tens of seconds on CPU
_NETS = [
    ("resnet10", 224, True), ("resnetbc14b", 224, True), ("resnet34", 224, True), ("resnet152b", 224, False),
    ("mobilenet_wd2", 224, True), ("mobilenet_w3d4", 224, True),
    ("mobilenetv3_small_w7d20", 224, True), ("mobilenetv3_large_w5d4", 224, True),
    ("efficientnet_b1", 240, True), ("efficientnet_b3b", 300, True), ("efficientnet_b5c", 456, False),
    ("resnext50_32x4d", 224, True), ("resnext101_64x4d", 224, False), ("seresnet18", 224, True), ("seresnext101_32x4d", 224, False),
    ("preresnet34", 224, True), ("preresnetbc26b", 224, True), ("preresnet269b", 224, False), ("sepreresnet50b", 224, True),
    ("densenet169", 224, True), ("densenet161", 224, False),
]


@pytest.mark.parametrize("name,size,check", _NETS, ids=[n for n, _, _ in _NETS])
def test_registry_model_runs_and_matches_oracle(name, size, check, cuda_device):
    import pytorchcv_amd
    from pytorchcv_amd.model_provider import get_model
    net = get_model(name).eval()
    assert tuple(net.in_size) == (size, size)
    sd = util.synth_state_dict(net.state_dict(), seed=5)
    net.load_state_dict(sd, strict=True)
    x = util.synth_input(2, 3, size, size, seed=9)
    net = pytorchcv_amd.set_compute_dtype(net.to(cuda_device), "fp32" if check else "bf16")
    with torch.no_grad():
        y = net(x.to(cuda_device))
    torch.cuda.synchronize()
    y = y.cpu()
    assert y.shape == (2, 1000) and bool(torch.isfinite(y).all())
    if check:
        ref = refnet.forward(name, {k: v.float() for k, v in sd.items()}, x)
        err = float((y - ref).abs().max())
        # uncalibrated random weights let activations grow to 1e2..1e3 and make the nets ill-conditioned: the CPU oracle's own
        # fp32-vs-fp64 distance on efficientnet_b3b is 6e-3 of max|logit|, so 5e-3 is what fp32 can be held to here (a wrong
        # padding or channel order shows up as O(1) relative error)
        assert err <= 5e-3 * max(1.0, float(ref.abs().max())), (name, err)


def test_unsupported_width_variants_are_refused_before_any_launch(cuda_device):
    """They construct (state_dict / parameter count like the reference's) but their forward names the offending layer."""
    from pytorchcv_amd.model_provider import get_model
    for name in ("mobilenetv2_wd4", "mobilenetv2_wd2", "mobilenetv2_w3d4"):
        net = get_model(name).eval().to(cuda_device)
        with pytest.raises(NotImplementedError, match="multiples of 8"):
            net(torch.zeros(1, 3, 224, 224, device=cuda_device))
