"""
    ORACLE - test infrastructure, not product code: ctypes view of oracle/cref.c (plain-C per-op restatement).
"""

__all__ = ['build', 'lib', 'conv_block_c', 'maxpool2d_c', 'avgpool2d_c', 'linear_c', 'se_gate_c']

import os
import ctypes
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "libcref.so"])


def lib():
    """libcref.so, or the library named by ORACLE_CREF_LIB (the ASan + UBSan build, tests/test_host_sanitizers.py)."""
    global _LIB
    if _LIB is None:
        path = os.environ.get("ORACLE_CREF_LIB") or os.path.join(_HERE, "libcref.so")
        if not os.path.exists(path):
            build()
        _LIB = ctypes.CDLL(path)
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def _f32(a):
    return None if a is None else np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def conv_block_c(x, w, bias=None, bn=None, eps=1e-5, stride=1, padding=0, dilation=1, groups=1, act=None,
                 residual=None, post_act=None):
    """ConvBlock (conv.py:278-286) + optional residual/post-activation, plain C. padding: int or (l, r, t, b)."""
    x, w, bias = _f32(x), _f32(w), _f32(bias)
    N, C, H, W = x.shape
    O, _, kh, kw = w.shape
    if isinstance(padding, (list, tuple)):
        pl, pr, pt, pb = padding
    else:
        pl = pr = pt = pb = int(padding)
    Ho = (H + pt + pb - dilation * (kh - 1) - 1) // stride + 1
    Wo = (W + pl + pr - dilation * (kw - 1) - 1) // stride + 1
    y = np.empty((N, O, Ho, Wo), dtype=np.float32)
    L = lib()
    rc = L.cref_conv2d(_p(x), _p(w), _p(bias), _p(y), N, C, H, W, O, kh, kw, stride, stride,
                       pt, pl, pb, pr, dilation, dilation, groups)
    assert rc == 0
    if bn is not None:
        g, b, m, v = [_f32(t) for t in bn]
        L.cref_bn_eval(_p(y), _p(g), _p(b), _p(m), _p(v), ctypes.c_float(eps), N, O, ctypes.c_long(Ho * Wo))
    code = {None: 0, "relu": 1, "relu6": 2, "sigmoid": 3, "swish": 4, "hsigmoid": 5, "hswish": 6}
    L.cref_act(_p(y), ctypes.c_long(y.size), code[act])
    if residual is not None:
        r = _f32(residual)
        L.cref_add(_p(y), _p(r), ctypes.c_long(y.size))
    L.cref_act(_p(y), ctypes.c_long(y.size), code[post_act])
    return y


def maxpool2d_c(x, k, s, p):
    x = _f32(x)
    N, C, H, W = x.shape
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    y = np.empty((N, C, Ho, Wo), dtype=np.float32)
    lib().cref_maxpool2d(_p(x), _p(y), N, C, H, W, k, s, p)
    return y


def avgpool2d_c(x, k, s):
    x = _f32(x)
    N, C, H, W = x.shape
    Ho, Wo = (H - k) // s + 1, (W - k) // s + 1
    y = np.empty((N, C, Ho, Wo), dtype=np.float32)
    lib().cref_avgpool2d(_p(x), _p(y), N, C, H, W, k, s)
    return y


def linear_c(x, w, b=None):
    x, w, b = _f32(x), _f32(w), _f32(b)
    y = np.empty((x.shape[0], w.shape[0]), dtype=np.float32)
    lib().cref_linear(_p(x), _p(w), _p(b), _p(y), x.shape[0], x.shape[1], w.shape[0])
    return y


def se_gate_c(x, w1, b1, w2, b2):
    x = _f32(x)
    N, C, H, W = x.shape
    w1, b1, w2, b2 = _f32(w1).reshape(-1, C), _f32(b1), _f32(w2), _f32(b2)
    M = w1.shape[0]
    w2 = w2.reshape(C, M)
    g = np.empty((N, C), dtype=np.float32)
    t1 = np.empty(C, dtype=np.float32)
    t2 = np.empty(M, dtype=np.float32)
    lib().cref_se_gate(_p(x), _p(w1), _p(b1), _p(w2), _p(b2), _p(g), _p(t1), _p(t2), N, C, ctypes.c_long(H * W), M)
    return g
