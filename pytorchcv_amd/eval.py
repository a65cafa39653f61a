"""
    Evaluation harness for the ImageNet-1K classifiers: the preprocessing the reference's pretrained weights assume
    ("ordinary normalization" of a centre crop: README.md:12-13; `img_size`, `img_scale = 0.875` of
    models/common/model_metainfos.csv:1) and top-k error as its README tables quote it. The reference keeps these scripts
    out of tree (imgclsmob); here they sit next to the hot path because the input format is part of it: decoded uint8 frames
    go through ONE kernel (crop + normalise + NHWC4 layout + cast, pcv_preprocess_u8) straight into the stem convolution.

    Resizing a decoded image to `resize_size(...)` (shorter side, bilinear) is the decoder's business (PIL / DALI / rocJPEG);
    this module starts from frames that already have that size.
"""

__all__ = ['IMAGENET_MEAN', 'IMAGENET_STD', 'resize_size', 'center_crop_box', 'preprocess_u8', 'topk_errors', 'evaluate']

import math
import torch
from . import engine, _lib

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def resize_size(img_size: int = 224, img_scale: float = 0.875) -> int:
    """Shorter-side size before the centre crop: ceil(img_size / img_scale) = 256 for 224 / 0.875."""
    return int(math.ceil(float(img_size) / img_scale))


def center_crop_box(height: int, width: int, img_size: int = 224):
    """(top, left) of the img_size x img_size centre crop, torchvision's rounding."""
    if height < img_size or width < img_size:
        raise ValueError("frame {}x{} is smaller than the {} crop".format(height, width, img_size))
    return int(round((height - img_size) / 2.0)), int(round((width - img_size) / 2.0))


def preprocess_u8(frames: torch.Tensor, img_size: int = 224, dtype: str = "bf16", mean=IMAGENET_MEAN, std=IMAGENET_STD) -> engine.NHWC:
    """uint8 [N, Hs, Ws, C<=4] device tensor -> the stem's input handle (centre crop, normalise, NHWC4, cast)."""
    if frames.dtype != torch.uint8 or frames.dim() != 4 or frames.shape[3] > 4:
        raise TypeError("expected a uint8 tensor [N, H, W, C <= 4]")
    frames = frames.contiguous()
    n, hs, ws, c = frames.shape
    if len(mean) < c or len(std) < c:
        raise ValueError("mean/std need one value per channel")
    top, left = center_crop_box(hs, ws, img_size)
    code, tdt = engine.DTYPES[dtype]
    wp = (img_size + 1) // 2 * 2
    dev = frames.device
    y = torch.empty((n, img_size, wp, 4), dtype=tdt, device=dev)
    m = torch.tensor(list(mean)[:c] + [0.0] * (4 - c), dtype=torch.float32, device=dev)
    s = torch.tensor([1.0 / v for v in list(std)[:c]] + [0.0] * (4 - c), dtype=torch.float32, device=dev)
    ctx = engine._ctx(dev)
    _lib.check(_lib.lib().pcv_preprocess_u8(ctx, engine._ptr(frames), engine._ptr(y), n, hs, ws, c, top, left, img_size, img_size,
                                            wp, engine._ptr(m), engine._ptr(s), code, engine._stream(dev)), ctx)
    return engine.NHWC(y, n, img_size, img_size, c, wpitch=wp, cpitch=4)


def topk_errors(logits: torch.Tensor, labels: torch.Tensor, ks=(1, 5)):
    """Number of samples whose label is NOT among the k largest logits, for each k."""
    top = logits.topk(max(ks), dim=1).indices
    hit = top.eq(labels.view(-1, 1))
    return [int(labels.numel() - hit[:, :k].any(dim=1).sum()) for k in ks]


def evaluate(net, batches, img_size: int = 224, ks=(1, 5)):
    """`batches`: iterable of (uint8 frames [N, Hs, Ws, 3] on the net's device, int64 labels [N]).
    Returns {"top1_err": ..., "top5_err": ..., "n": ...} in the README's convention (error rates in %)."""
    dtype = engine.compute_dtype_of(net)
    wrong = [0] * len(ks)
    total = 0
    with torch.no_grad():
        for frames, labels in batches:
            logits = net(preprocess_u8(frames, img_size=img_size, dtype=dtype))
            for i, w in enumerate(topk_errors(logits, labels.to(logits.device), ks)):
                wrong[i] += w
            total += int(labels.numel())
    out = {"n": total}
    for k, w in zip(ks, wrong):
        out["top{}_err".format(k)] = 100.0 * w / max(total, 1)
    return out
