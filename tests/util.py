"""Shared helpers for the parity tests: fixture loading and block/model construction on both sides."""

import os
import json
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")

from pytorchcv_amd.synth import synth_state_dict, synth_input, hash_normal  # noqa: E402
from cases import BLOCK_CASES, MODELS, F4_CASES, build_f4_block  # noqa: E402,F401

import contextlib  # noqa: E402


# defaults of the pcv_set_tuning switches the tests flip (restored on exit, also when the body raises)
_TUNING_DEFAULTS = {"max_blocks": 0, "d3x3": -1, "d3w": -1, "d3c": -1, "d3k": -1, "d3i": -1, "d1i": -1, "p1r": -1, "tile": -1, "wstat": 1, "pair_pb": 2, "wpair": 3, "persist": 1, "head": 1, "mbw": 1, "mbr": 1, "mbr_xl": 1, "d1x1": -1, "gconvr": 1, "mbw_wide": 1, "stem32": 1, "dbg": 0}


@contextlib.contextmanager
def tuning(device=0, **switches):
    """`with util.tuning(max_blocks=8): ...` - set library tuning switches for the body, restore their defaults after it.
    `max_blocks=n` caps every persistent grid at n blocks so that a small fixture walks several tiles per block."""
    from pytorchcv_amd import _lib
    ctx = _lib.ctx_for(device)
    L = _lib.lib()
    try:
        for k, v in switches.items():
            _lib.check(L.pcv_set_tuning(ctx, k.encode(), int(v)), ctx)
        yield
    finally:
        for k in switches:
            _lib.check(L.pcv_set_tuning(ctx, k.encode(), _TUNING_DEFAULTS[k]), ctx)


_blocks_npz = None
_blocks_meta = None


def blocks_golden():
    global _blocks_npz, _blocks_meta
    if _blocks_npz is None:
        _blocks_npz = np.load(os.path.join(GOLDEN, "blocks.npz"))
        with open(os.path.join(GOLDEN, "blocks.json")) as f:
            _blocks_meta = json.load(f)
    return _blocks_npz, _blocks_meta


def template_from_manifest(manifest: dict) -> dict:
    return {k: torch.zeros(shape, dtype=getattr(torch, dt)) for k, (shape, dt) in manifest.items()}


def block_state_and_input(case):
    _, meta = blocks_golden()
    m = meta[case["name"]]
    sd = synth_state_dict(template_from_manifest(m["manifest"]), seed=m["weight_seed"])
    n, c, h, w = case["x"]
    x = synth_input(n, c, h, w, seed=m["input_seed"])
    return sd, x


def block_golden(case) -> torch.Tensor:
    npz, _ = blocks_golden()
    return torch.from_numpy(npz[case["name"]])


def build_block(case):
    """The pytorchcv_amd counterpart of the reference block named by the case."""
    from pytorchcv_amd.models.common import conv as C
    from pytorchcv_amd.models.common.att import SEBlock
    from pytorchcv_amd.models.common.activ import lambda_relu6
    from pytorchcv_amd.models.resnet import ResUnit, ResInitBlock
    from pytorchcv_amd.models.mobilenetv2 import LinearBottleneck
    from pytorchcv_amd.models.resnext import ResNeXtUnit
    from pytorchcv_amd.models.seresnet import SEResUnit
    from pytorchcv_amd.models.mobilenetv3 import MobileNetV3Unit
    from pytorchcv_amd.models.efficientnet import EffiInitBlock, EffiDwsConvUnit, EffiInvResUnit
    from pytorchcv_amd.models.common.activ import lambda_swish
    from pytorchcv_amd.models.common.norm import lambda_batchnorm2d
    from pytorchcv_amd.models.preresnet import PreResUnit, PreResInitBlock, PreResActivation
    from pytorchcv_amd.models.densenet import DenseUnit, TransitionBlock
    from pytorchcv_amd.models.shufflenetv2 import ShuffleUnit, ShuffleInitBlock
    kind, kw = case["kind"], dict(case["kwargs"])
    if kind == "LinearBottleneck":
        kw["activation"] = lambda_relu6()
    if kind.startswith("Effi"):
        kw["normalization"] = lambda_batchnorm2d(eps=kw.pop("bn_eps"))
        kw["activation"] = lambda_swish()
    ctor = {"ConvBlock": C.ConvBlock, "conv1x1_block": C.conv1x1_block, "conv3x3_block": C.conv3x3_block,
            "conv7x7_block": C.conv7x7_block, "dwconv3x3_block": C.dwconv3x3_block, "dwconv5x5_block": C.dwconv5x5_block,
            "SEBlock": SEBlock, "ResUnit": ResUnit, "ResInitBlock": ResInitBlock, "LinearBottleneck": LinearBottleneck,
            "ResNeXtUnit": ResNeXtUnit, "SEResUnit": SEResUnit, "MobileNetV3Unit": MobileNetV3Unit,
            "EffiInitBlock": EffiInitBlock, "EffiDwsConvUnit": EffiDwsConvUnit, "EffiInvResUnit": EffiInvResUnit,
            "pre_conv3x3_block": C.pre_conv3x3_block, "pre_conv1x1_block": C.pre_conv1x1_block, "PreResUnit": PreResUnit,
            "PreResInitBlock": PreResInitBlock, "PreResActivation": PreResActivation, "DenseUnit": DenseUnit,
            "TransitionBlock": TransitionBlock, "ShuffleUnit": ShuffleUnit, "ShuffleInitBlock": ShuffleInitBlock}[kind]
    return ctor(**kw).eval()


def model_manifest(name):
    with open(os.path.join(GOLDEN, "manifest_{}.json".format(name))) as f:
        return json.load(f)


def model_calib(name):
    with open(os.path.join(GOLDEN, "calib_{}.json".format(name))) as f:
        return {k: tuple(v) for k, v in json.load(f).items()}


_state_cache = {}       # name -> state dict; the integer-hash generator is bit-stable but slow (VGG-11: 133 M values, ~1 min)


def model_state(name, template=None):
    """The fixture state dict of a golden model (same values whichever template - the manifest's or a live net's - names the keys).
    Cached per process (last three models); callers load it into modules, nobody writes into it."""
    if name not in _state_cache:
        if template is None:
            template = template_from_manifest(model_manifest(name)["keys"])
        while len(_state_cache) >= 3:
            _state_cache.pop(next(iter(_state_cache)))
        _state_cache[name] = synth_state_dict(template, seed=1234, calib=model_calib(name))
    return dict(_state_cache[name])


def model_golden(name):
    z = np.load(os.path.join(GOLDEN, "logits_{}.npz".format(name)))
    return torch.from_numpy(z["logits"]), [int(i) for i in z["image_ids"]]


def model_digests(name):
    with open(os.path.join(GOLDEN, "digests_{}.json".format(name))) as f:
        return json.load(f)


def images(ids):
    return torch.stack([torch.from_numpy(hash_normal(0, 0x1A9E0000 + i, 3 * 224 * 224).astype(np.float32)).view(3, 224, 224)
                        for i in ids])


def digest(t: torch.Tensor, sample=64):
    f = t.double().flatten()
    idx = torch.linspace(0, f.numel() - 1, sample).long()
    return dict(shape=list(t.shape), sum=float(f.sum()), sumsq=float((f * f).sum()),
                samples=[float(v) for v in t.flatten()[idx].float()])


_f4 = None


def f4_golden(name):
    """(state dict, input, golden output of the imported reference) of a rank-4 block case (tests/golden/make_golden_f4.py)."""
    global _f4
    if _f4 is None:
        with open(os.path.join(GOLDEN, "blocks_f4.json")) as f:
            _f4 = (np.load(os.path.join(GOLDEN, "blocks_f4.npz")), json.load(f))
    npz, meta = _f4
    m = meta[name]
    sd = synth_state_dict(template_from_manifest(m["manifest"]), seed=m["weight_seed"]) if m["manifest"] else {}
    x = synth_input(*F4_CASES[name], seed=m["input_seed"])
    return sd, x, torch.from_numpy(npz[name])


def build_f4(name):
    """The pytorchcv_amd counterpart of a rank-4 block case."""
    import types
    import torch.nn as nn
    from pytorchcv_amd.models.common import arch, tutti, conv
    from pytorchcv_amd.models._tail import MaxPool2dNHWC
    ns = types.SimpleNamespace(Concurrent=arch.Concurrent, SequentialConcurrent=arch.SequentialConcurrent,
                               NormActivation=tutti.NormActivation, InterpolationBlock=tutti.InterpolationBlock,
                               ChannelShuffle=tutti.ChannelShuffle, conv1x1_block=conv.conv1x1_block,
                               conv3x3_block=conv.conv3x3_block, Sequential=nn.Sequential,
                               MaxPool=lambda: MaxPool2dNHWC(kernel_size=3, stride=1, padding=1))
    return build_f4_block(name, ns).eval()
