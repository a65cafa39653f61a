"""
    Bit-stable synthetic weights and inputs for the conv-net inference path.

    There is no network in the build pipeline, so pretrained `.pth` files (reference:
    pytorchcv/models/common/model_store.py:140-192) cannot be fetched. Parity fixtures and the bench
    therefore use weights produced here: a counter-hash RNG written with integer arithmetic only
    (splitmix64), so the same (seed, parameter-name, index) triple gives the same fp32 value on every
    machine and every numpy/torch version. No libm call (log/cos) is on the path: normal-like
    variates are Irwin-Hall(4) sums.

    Distribution: conv weights U(+-sqrt(6/fan_in)) as
    `nn.init.kaiming_uniform_` gives in reference resnet.py:326-331; BN gamma U(0.5,1.5),
    beta 0.2*N(0,1); BN running statistics from a small per-layer calibration table
    (tests/golden/calib_<model>.json) so that activations stay O(1) through the whole net.
"""

__all__ = ['hash_uniform', 'hash_normal', 'synth_input', 'synth_state_dict', 'name_stream']

import zlib
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
_GOLD = np.uint64(0x9E3779B97F4A7C15)
_C1 = np.uint64(0xBF58476D1CE4E5B9)
_C2 = np.uint64(0x94D049BB133111EB)
_SQRT3 = 1.7320508075688772


def _splitmix64(z: np.ndarray) -> np.ndarray:
    with np.errstate(over='ignore'):
        z = (z + _GOLD) & _M64
        z = ((z ^ (z >> np.uint64(30))) * _C1) & _M64
        z = ((z ^ (z >> np.uint64(27))) * _C2) & _M64
        z = z ^ (z >> np.uint64(31))
    return z


def name_stream(name: str) -> int:
    """Stable 32-bit stream id of a parameter name."""
    return zlib.crc32(name.encode("utf-8")) & 0xFFFFFFFF


_CHUNK = 1 << 16        # elements per pass: the integer temporaries of a chunk (512 KB each) stay in cache and are recycled by the allocator


def _raw(seed: int, stream: int, lane: int, n: int, start: int = 0) -> np.ndarray:
    idx = np.arange(start, start + n, dtype=np.uint64)
    with np.errstate(over='ignore'):
        key = (np.uint64(seed) * np.uint64(0xD1342543DE82EF95)
               + np.uint64(stream) * np.uint64(0xA0761D6478BD642F)
               + np.uint64(lane) * np.uint64(0xE7037ED1A0B428DB)) & _M64
        z = _splitmix64((idx * np.uint64(0x2545F4914F6CDD1D) + key) & _M64)
    return z


def hash_uniform(seed: int, stream: int, n: int, lane: int = 0) -> np.ndarray:
    """n float64 values in [0, 1) with 24 significant bits (exact in fp32). Evaluated chunk by chunk (same values: element i
    depends on (seed, stream, lane, i) only) - a 100 M-element classifier matrix in one piece streams ~20 GB of temporaries."""
    out = np.empty(n, dtype=np.float64)
    for a in range(0, n, _CHUNK):
        m = min(_CHUNK, n - a)
        z = _raw(seed, stream, lane, m, start=a)
        out[a:a + m] = (z >> np.uint64(40)).astype(np.float64) * (1.0 / 16777216.0)
    return out


def _uniform_pm_f32(seed: int, stream: int, n: int, bound: float) -> np.ndarray:
    """float32((2 u - 1) * bound) for u = hash_uniform(seed, stream, n), produced chunk by chunk straight into the fp32 array (the
    same values as the float64 expression followed by astype: elementwise; no gigabyte-sized float64 intermediates)."""
    out = np.empty(n, dtype=np.float32)
    for a in range(0, n, _CHUNK):
        m = min(_CHUNK, n - a)
        u = (_raw(seed, stream, 0, m, start=a) >> np.uint64(40)).astype(np.float64) * (1.0 / 16777216.0)
        out[a:a + m] = ((u * 2.0 - 1.0) * bound).astype(np.float32)
    return out


def hash_normal(seed: int, stream: int, n: int, lane: int = 0) -> np.ndarray:
    """n float64 values, zero mean, unit variance (Irwin-Hall sum of 4 uniforms)."""
    s = np.zeros(n, dtype=np.float64)
    for j in range(4):
        s += hash_uniform(seed, stream, n, lane=lane * 4 + j + 1)
    return (s - 2.0) * _SQRT3


def synth_input(batch: int, channels: int = 3, height: int = 224, width: int = 224, seed: int = 0):
    """Seeded N(0,1)-like NCHW fp32 batch as a torch tensor (image i depends only on (seed, i))."""
    import torch
    per = channels * height * width
    out = np.empty((batch, per), dtype=np.float32)
    for i in range(batch):
        out[i] = hash_normal(seed, 0x1A9E0000 + i, per).astype(np.float32)
    return torch.from_numpy(out.reshape(batch, channels, height, width))


def synth_state_dict(template: dict, seed: int = 1234, calib: dict | None = None) -> dict:
    """
    Build a state_dict with the keys/shapes/dtypes of `template` (a `net.state_dict()`).

    calib maps a BatchNorm prefix (e.g. 'features.init_block.conv.bn.') to (mean, var) of the
    layer's pre-BN activations; missing entries default to (0, 1).
    """
    import torch
    calib = calib or {}
    keys = list(template.keys())
    keyset = set(keys)
    out = {}
    for k in keys:
        t = template[k]
        shape = tuple(t.shape)
        n = int(np.prod(shape)) if len(shape) > 0 else 1
        st = name_stream(k)
        prefix = k[:k.rfind(".") + 1]
        leaf = k[k.rfind(".") + 1:]
        is_bn = (prefix + "running_mean") in keyset
        if leaf == "num_batches_tracked":
            out[k] = torch.zeros(shape, dtype=t.dtype)
            continue
        if is_bn:
            m0, v0 = calib.get(prefix, (0.0, 1.0))
            if leaf == "weight":
                v = 0.5 + hash_uniform(seed, st, n)
            elif leaf == "bias":
                v = 0.2 * hash_normal(seed, st, n)
            elif leaf == "running_mean":
                v = float(m0) + 0.1 * float(np.sqrt(np.float64(v0))) * hash_normal(seed, st, n)
            elif leaf == "running_var":
                v = float(v0) * (0.8 + 0.45 * hash_uniform(seed, st, n))
            else:
                raise ValueError("unexpected BatchNorm entry: {}".format(k))
        elif leaf == "weight" and len(shape) == 4:
            fan_in = shape[1] * shape[2] * shape[3]
            bound = float(np.sqrt(np.float64(6.0) / np.float64(fan_in)))
            v = _uniform_pm_f32(seed, st, n, bound)
        elif leaf == "weight" and len(shape) == 2:
            bound = float(1.0 / np.sqrt(np.float64(shape[1])))
            v = _uniform_pm_f32(seed, st, n, bound)
        elif leaf == "bias":
            v = (hash_uniform(seed, st, n) * 2.0 - 1.0) * 0.1
        else:
            raise ValueError("no synthetic rule for state_dict entry: {} {}".format(k, shape))
        out[k] = torch.from_numpy(v.astype(np.float32).reshape(shape)).to(t.dtype)
    return out
