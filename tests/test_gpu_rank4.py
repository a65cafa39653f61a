"""GPU parity of the SURVEY 8(f) rank-4 blocks - Concurrent (cat / sum), SequentialConcurrent, NormActivation, InterpolationBlock,
ChannelShuffle - through the C ABI: fp32 against the golden outputs of the reference's own classes (<= 1e-3), 16-bit against the
quantisation-matched oracle; the merge-free second forward of a Concurrent (branches writing their channel slices / taking the
running sum as residual) is bit-identical to the first, copy-merged one."""

import pytest
import torch
import util
from oracle import refblocks

pytestmark = pytest.mark.gpu
NAMES = sorted(util.F4_CASES)


def _run(name, dtype, dev, times=1):
    import pytorchcv_amd
    sd, x, g = util.f4_golden(name)
    blk = util.build_f4(name)
    blk.load_state_dict(sd, strict=True)
    blk = pytorchcv_amd.set_compute_dtype(blk.to(dev), dtype)
    ys = []
    with torch.no_grad():
        for _ in range(times):
            ys.append(blk(x.to(dev)).cpu())
    torch.cuda.synchronize()
    return sd, x, g, ys


@pytest.mark.parametrize("name", NAMES)
def test_rank4_block_fp32_matches_reference_golden(name, cuda_device):
    _, _, g, (y,) = _run(name, "fp32", cuda_device)
    assert y.shape == g.shape and y.dtype == torch.float32
    err = float((y - g).abs().max())
    assert err <= 1e-3, "fp32 max-abs error {:.3e}".format(err)


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("name", NAMES)
def test_rank4_block_16bit_matches_oracle(name, dtype, cuda_device):
    sd, x, g, (y,) = _run(name, dtype, cuda_device)
    ref = refblocks.f4_block_forward(name, sd, x, quant=dtype)
    d = (y - ref).abs()
    assert bool((d <= 1e-2 * torch.clamp(ref.abs(), min=1.0)).all()), "vs quantisation-matched oracle: max |d| {:.3e}".format(float(d.max()))
    dg = (y - g).abs()
    tol = (4e-2, 2.0 ** -6) if dtype == "bf16" else (1e-2, 2.0 ** -9)
    assert bool((dg <= tol[0] + tol[1] * g.abs()).all()), "vs fp32 golden: max |d| {:.3e}".format(float(dg.max()))


@pytest.mark.parametrize("name", ["concurrent_cat", "concurrent_cat_pool", "concurrent_sum"])
def test_concurrent_merge_free_forward_equals_copy_merge(name, cuda_device):
    """First forward of a shape: branches run unmerged and are copied together (and the shapes recorded). Later forwards: a
    branch that ends in a ConvBlock writes its slice of the merged tensor from the convolution epilogue. Same bits."""
    from pytorchcv_amd.models.common.arch import Concurrent
    for dtype in ("fp32", "bf16"):
        _, _, _, ys = _run(name, dtype, cuda_device, times=3)
        assert torch.equal(ys[0], ys[1]) and torch.equal(ys[1], ys[2])
    blk = util.build_f4(name)
    assert isinstance(blk, Concurrent)
