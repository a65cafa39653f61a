"""Evaluation harness (pytorchcv_amd/eval.py): host-side rules on CPU, the fused uint8 preprocessing kernel on the GPU."""

import pytest
import torch
import util


def test_resize_and_crop_rules():
    from pytorchcv_amd import eval as ev
    assert ev.resize_size(224, 0.875) == 256                     # model_metainfos.csv: img_size 224, img_scale 0.875
    assert ev.resize_size(299, 0.875) == 342
    assert ev.center_crop_box(256, 256, 224) == (16, 16)
    assert ev.center_crop_box(256, 341, 224) == (16, 58)         # torchvision CenterCrop rounding
    with pytest.raises(ValueError):
        ev.center_crop_box(200, 300, 224)


def test_topk_errors():
    from pytorchcv_amd import eval as ev
    logits = torch.tensor([[0.1, 0.9, 0.0, 0.3, 0.2, 0.05], [0.9, 0.1, 0.2, 0.3, 0.4, 0.5], [0.0, 0.1, 0.2, 0.3, 0.4, 0.5]])
    labels = torch.tensor([1, 1, 0])
    assert ev.topk_errors(logits, labels, ks=(1, 5)) == [2, 2]   # sample 0 right; 1: label rank 6; 2: label rank 6
    assert ev.topk_errors(logits, torch.tensor([1, 0, 5]), ks=(1, 5)) == [0, 0]


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["fp32", "bf16", "fp16"])
def test_preprocess_u8_matches_torch(dtype, cuda_device):
    from pytorchcv_amd import eval as ev
    g = torch.Generator().manual_seed(3)
    frames = torch.randint(0, 256, (3, 256, 341, 3), generator=g, dtype=torch.uint8)
    a = ev.preprocess_u8(frames.to(cuda_device), img_size=224, dtype=dtype)
    torch.cuda.synchronize()
    assert (a.N, a.H, a.W, a.C, a.cpitch, a.wpitch) == (3, 224, 224, 3, 4, 224)
    top, left = ev.center_crop_box(256, 341, 224)
    crop = frames[:, top:top + 224, left:left + 224, :].float() / 255.0
    ref = (crop - torch.tensor(ev.IMAGENET_MEAN)) / torch.tensor(ev.IMAGENET_STD)
    got = a.t.float().cpu()
    assert float(got[..., 3].abs().max()) == 0.0                 # pad channel
    tol = {"fp32": 1e-6, "bf16": 2.0 ** -7, "fp16": 2.0 ** -10}[dtype]
    assert float((got[..., :3] - ref).abs().max()) <= tol * max(1.0, float(ref.abs().max()))


@pytest.mark.gpu
def test_evaluate_runs_from_uint8_frames(cuda_device):
    """uint8 frames -> logits through the fused preprocessing equals the fp32-NCHW entry of the same net."""
    import pytorchcv_amd
    from pytorchcv_amd import eval as ev
    from pytorchcv_amd.model_provider import get_model
    net = get_model("resnet18").eval()
    net.load_state_dict(util.model_state("resnet18", net.state_dict()), strict=True)
    net = pytorchcv_amd.set_compute_dtype(net.to(cuda_device), "fp32")
    g = torch.Generator().manual_seed(4)
    frames = torch.randint(0, 256, (4, 256, 256, 3), generator=g, dtype=torch.uint8)
    labels = torch.tensor([1, 2, 3, 4])
    with torch.no_grad():
        y_u8 = net(ev.preprocess_u8(frames.to(cuda_device), dtype="fp32"))
        crop = frames[:, 16:240, 16:240, :].float() / 255.0
        x = ((crop - torch.tensor(ev.IMAGENET_MEAN)) / torch.tensor(ev.IMAGENET_STD)).permute(0, 3, 1, 2).contiguous()
        y_ref = net(x.to(cuda_device))
    assert float((y_u8 - y_ref).abs().max()) <= 1e-4 * max(1.0, float(y_ref.abs().max()))
    res = ev.evaluate(net, [(frames.to(cuda_device), labels.to(cuda_device))])
    assert res["n"] == 4 and 0.0 <= res["top1_err"] <= 100.0 and res["top5_err"] <= res["top1_err"]
