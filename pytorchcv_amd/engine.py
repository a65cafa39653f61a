"""
    Host-side plumbing between the pytorchcv-style module tree and the HIP kernels: NHWC activation handles, weight
    pre-packing (BN folding, K-major MFMA-ordered weights) cached per module, and one function per C-ABI hot-path call.
    PyTorch is used for device memory and streams only.
"""

__all__ = ['NHWC', 'DTYPES', 'default_dtype', 'set_compute_dtype', 'compute_dtype_of', 'Fp16Guard', 'fp16_overflow_count', 'from_nchw', 'to_nchw', 'ConvRunner',
           'BnActRunner', 'maxpool2d', 'avgpool2d', 'global_avgpool', 'se_forward', 'channel_slice', 'cat_shuffle2', 'act_code',
           'boundary', 'round8', 'channel_concat_into', 'interpolate', 'add']

import os
import ctypes
import torch
import torch.nn as nn
import torch.nn.functional as F
from . import _lib
from ._lib import ConvDesc

DTYPES = {"fp32": (0, torch.float32), "bf16": (1, torch.bfloat16), "fp16": (2, torch.float16)}
_CODE_OF_TORCH = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}
_NAME_OF_TORCH = {torch.float32: "fp32", torch.bfloat16: "bf16", torch.float16: "fp16"}


def default_dtype() -> str:
    """Process-wide storage / MFMA type of the hot path (env PCV_AMD_DTYPE): "auto" (default), "bf16", "fp16" or "fp32"."""
    d = os.environ.get("PCV_AMD_DTYPE", "auto")
    if d not in DTYPES and d != "auto":
        raise ValueError("PCV_AMD_DTYPE must be one of {}".format(sorted(DTYPES) + ["auto"]))
    return d


def set_compute_dtype(net: nn.Module, dtype: str) -> nn.Module:
    """Select the storage/MFMA type of the hot path for `net`: "auto" (default: the family's 16-bit mode, see
    `compute_dtype_of`), "bf16", "fp16" or "fp32"."""
    if dtype not in DTYPES and dtype != "auto":
        raise ValueError("dtype must be one of {}".format(sorted(DTYPES) + ["auto"]))
    for m in net.modules():
        m._pcv_dtype = dtype
    return net


def stamp_family_dtype(net: nn.Module) -> nn.Module:
    """Give every sub-module of `net` the 16-bit mode its family declares (`pcv_16bit` on the net class), so that "auto" resolves to the
    SAME type whether the whole net or one of its parts (`net.features(x)`, a unit, a block) is called with an NCHW tensor. Called at
    the end of the family's constructor."""
    mode = getattr(net, "pcv_16bit", None)
    if mode is not None:
        for m in net.modules():
            if m is not net:
                m.pcv_16bit = mode
    return net


def compute_dtype_of(module: nn.Module) -> str:
    """The type `module` runs in. "auto" resolves to the module's own 16-bit mode: bf16, except for the net classes that declare
    `pcv_16bit = "fp16"` - the depthwise-separable families (MobileNetV2 / V3, EfficientNet), whose logits stay within the
    north-star 1e-2 of the reference's fp32 forward in fp16 (2e-3 .. 5e-3) but not in bf16 (1.3e-2 .. 2.2e-2: 8 mantissa bits on
    weights that multiply [0, 6]-bounded activations). Same MFMA rate, same bytes; fp16's narrower exponent
    range is covered by the range guard (`Fp16Guard`)."""
    d = getattr(module, "_pcv_dtype", None) or default_dtype()
    if d == "auto":
        d = getattr(module, "pcv_16bit", None) or "bf16"
    return d


class Fp16Guard(object):
    """fp16 range guard around one forward (include/pcv_amd.h, pcv_fp16_guard_begin / _end): every kernel counts the values it
    rounds beyond fp16's range; `finish(y)` - stream-ordered, no host synchronisation, capturable - overwrites the fp32 result `y`
    with NaN when the count moved during the forward. A no-op for bf16 / fp32."""
    __slots__ = ("slot", "device")

    def __init__(self, device, torch_dtype):
        self.slot = None
        self.device = device
        if torch_dtype == torch.float16:
            self.slot = torch.empty(1, dtype=torch.int32, device=device)
            ctx = _ctx(device)
            _lib.check(_lib.lib().pcv_fp16_guard_begin(ctx, _ptr(self.slot), _stream(device)), ctx)

    def finish(self, y: torch.Tensor) -> torch.Tensor:
        if self.slot is not None:
            if y.dtype != torch.float32 or not y.is_contiguous():
                raise RuntimeError("the fp16 guard poisons a contiguous fp32 result")
            ctx = _ctx(self.device)
            _lib.check(_lib.lib().pcv_fp16_guard_end(ctx, _ptr(self.slot), _ptr(y), y.numel(), _stream(self.device)), ctx)
        return y


def fp16_overflow_count(device) -> int:
    """How many threads of this device's context have rounded a value beyond fp16's range so far (synchronises the stream)."""
    ctx = _ctx(device)
    n = ctypes.c_uint(0)
    _lib.check(_lib.lib().pcv_fp16_overflow_count(ctx, ctypes.byref(n), _stream(device)), ctx)
    return int(n.value)


def round8(c: int) -> int:
    return (int(c) + 7) // 8 * 8


class NHWC(object):
    """
    An activation on the hot path: `t` is a contiguous device tensor [N, H, wpitch, cpitch]; (H, W, C) is the logical
    extent. The kernels move 16-byte channel chunks, so the physical channel count `cpitch` is C rounded up to a multiple
    of 8 (what the pad channels hold never reaches a logical channel: every weight that would read them is zero-padded).
    The network input is the exception: C <= 4 -> cpitch 4, W -> even (the stem kernel's layout).
    """
    __slots__ = ("t", "N", "H", "W", "C", "wpitch", "cpitch")

    def __init__(self, t, N, H, W, C, wpitch=None, cpitch=None):
        self.t, self.N, self.H, self.W, self.C = t, N, H, W, C
        self.wpitch = W if wpitch is None else wpitch
        self.cpitch = C if cpitch is None else cpitch

    @property
    def dtype(self):
        return self.t.dtype

    @property
    def device(self):
        return self.t.device

    @property
    def dense(self) -> bool:
        """Canonical internal layout: no row padding, channels padded to the next multiple of 8 only."""
        return self.wpitch == self.W and self.cpitch == round8(self.C)

    def size(self, dim=None):
        s = (self.N, self.C, self.H, self.W)        # reported NCHW-style, as callers of the reference expect
        return s if dim is None else s[dim]


class _ShapeOnly(object):
    """What ConvRunner.desc reads of its input, for a tensor that does not exist yet."""
    __slots__ = ("N", "H", "W", "C", "dtype", "wpitch", "cpitch")

    def __init__(self, N, H, W, C, dtype):
        self.N, self.H, self.W, self.C, self.dtype, self.wpitch, self.cpitch = N, H, W, C, dtype, W, round8(C)


def _stream(device) -> ctypes.c_void_p:
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _ctx(device):
    if device.type != "cuda":
        raise RuntimeError("pytorchcv_amd runs on MI355X (gfx950) only: got a tensor on '{}'. Move the model and the input "
                           "to a CUDA/HIP device; there is no CPU path in this package.".format(device))
    idx = device.index if device.index is not None else torch.cuda.current_device()
    return _lib.ctx_for(idx)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


class LazyNCHW(NHWC):
    """The network input as the caller gave it - fp32 NCHW - seen through the handle interface of its padded NHWC4 view. The stem
    kernel reads the fp32 planes itself (`ConvRunner._stem_from_nchw`, pcv_conv2d_nchw_stem_fused); anything else that asks for
    `.t` gets the converted tensor (pcv_nchw_to_nhwc, once)."""
    __slots__ = ("src", "_conv", "_dtype_name")

    def __init__(self, x: torch.Tensor, dtype: str):
        N, C, H, W = x.shape
        self.src, self._conv, self._dtype_name = x, None, dtype
        self.N, self.H, self.W, self.C = N, H, W, C
        self.wpitch, self.cpitch = (W + 1) // 2 * 2, 4

    @property
    def t(self):
        if self._conv is None:
            self._conv = from_nchw(self.src, self._dtype_name, stem=True).t
        return self._conv

    @property
    def materialized(self) -> bool:
        return self._conv is not None

    @property
    def dtype(self):
        return DTYPES[self._dtype_name][1]

    @property
    def device(self):
        return self.src.device


def network_input(x: torch.Tensor, dtype: str) -> NHWC:
    """Handle of a net's fp32 NCHW input: lazy (the stem kernel reads the planes directly) when the stem layout applies."""
    if x.dim() != 4:
        raise ValueError("expected a 4-D NCHW tensor")
    if x.shape[1] <= 3 and x.shape[3] % 4 == 0 and STEM_FROM_NCHW:
        return LazyNCHW(x.contiguous().float(), dtype)
    return from_nchw(x, dtype, stem=True)


def from_nchw(x: torch.Tensor, dtype: str, stem: bool = True) -> NHWC:
    """fp32 NCHW -> NHWC handle (pcv_nchw_to_nhwc). With `stem`, C <= 4 is padded to 4 channels and W to even."""
    if x.dim() != 4:
        raise ValueError("expected a 4-D NCHW tensor")
    x = x.contiguous().float()
    N, C, H, W = x.shape
    code, tdt = DTYPES[dtype]
    if C <= 4 and stem:
        cp, wp = 4, (W + 1) // 2 * 2
    else:
        cp, wp = (C + 7) // 8 * 8, W
    y = torch.empty((N, H, wp, cp), dtype=tdt, device=x.device)
    ctx = _ctx(x.device)
    _lib.check(_lib.lib().pcv_nchw_to_nhwc(ctx, _ptr(x), _ptr(y), N, C, H, W, cp, wp, code, _stream(x.device)), ctx)
    return NHWC(y, N, H, W, C, wpitch=wp, cpitch=cp)


def to_nchw(a: NHWC) -> torch.Tensor:
    """NHWC handle -> fp32 NCHW tensor (pcv_nhwc_to_nchw)."""
    if a.wpitch != a.W:
        raise RuntimeError("cannot convert a row-padded input handle back to NCHW")
    y = torch.empty((a.N, a.C, a.H, a.W), dtype=torch.float32, device=a.device)
    ctx = _ctx(a.device)
    _lib.check(_lib.lib().pcv_nhwc_to_nchw(ctx, _ptr(a.t), _ptr(y), a.N, a.C, a.H, a.W, a.cpitch, _CODE_OF_TORCH[a.dtype],
                                           _stream(a.device)), ctx)
    return y


def boundary(module: nn.Module, x, fn, stem: bool = False):
    """Run `fn` on the hot path. An `NHWC` handle passes straight through; an NCHW fp32 tensor (how the reference's
    blocks are called) is converted in and the result converted back, so a block is a drop-in on its own."""
    if isinstance(x, NHWC):
        return fn(x)
    if not torch.is_tensor(x):
        raise TypeError("expected a torch.Tensor or an NHWC handle")
    dtype = compute_dtype_of(module)
    guard = Fp16Guard(x.device, DTYPES[dtype][1])
    y = fn(from_nchw(x, dtype, stem=stem))
    if isinstance(y, tuple):                          # (output, pre-activated input) of the pre-activation blocks: the guard poisons both
        return tuple(guard.finish(to_nchw(v)) if isinstance(v, NHWC) else v for v in y)
    y = to_nchw(y) if isinstance(y, NHWC) else y
    return guard.finish(y) if torch.is_tensor(y) and y.dtype == torch.float32 else y


def act_code(activ) -> int:
    """pcv_act code of an activation module of common/activ.py (None -> 0)."""
    from .models.common.activ import Swish, HSigmoid, HSwish
    if activ is None:
        return 0
    if isinstance(activ, nn.ReLU):
        return 1
    if isinstance(activ, nn.ReLU6):
        return 2
    if isinstance(activ, nn.Sigmoid):
        return 3
    if isinstance(activ, Swish):
        return 4
    if isinstance(activ, HSigmoid):
        return 5
    if isinstance(activ, HSwish):
        return 6
    raise NotImplementedError("activation {} has no fused MI355X epilogue yet".format(type(activ).__name__))


def _pair(v):
    return (int(v[0]), int(v[1])) if isinstance(v, (tuple, list)) else (int(v), int(v))


class ConvRunner(object):
    """
    Packed state of one Conv2d (+ optional BatchNorm2d) for one (device, dtype, input channel pitch): weights in the
    kernel's layout and the folded fp32 scale/shift. Rebuilt whenever a source parameter changes (load_state_dict, .to()).
    """
    def __init__(self, conv: nn.Conv2d, bn, pad4=None):
        self.conv, self.bn, self.pad4 = conv, bn, pad4
        self._key = None
        self._foreign = False       # packed / scale / shift were RECEIVED (parallel.broadcast_packed_state): the local fp32 sources
                                    # are not what they were derived from, so nothing may be re-derived from them
        self.packed = self.scale = self.shift = None
        self.depthwise = (conv.groups > 1 and conv.groups == conv.in_channels == conv.out_channels)
        if 1 < conv.groups and not self.depthwise and (conv.in_channels % 8 or conv.out_channels % 8):
            raise NotImplementedError("grouped convolution with channel counts that are not multiples of 8")

    def _phys(self, x_cpitch: int, out_fp32: bool):
        """Physical (Cin, Cout, groups) the kernels run with: logical counts rounded up to multiples of 8, weights and BN
        constants zero-padded accordingly in prepare(). Exceptions: the padded network input of the stem (cpitch 4) keeps the
        logical Cin, and the fp32 classifier output keeps its logical width (ragged epilogue)."""
        c = self.conv
        cin = c.in_channels if x_cpitch == 4 and c.in_channels <= 4 else round8(c.in_channels)
        cout = c.out_channels if out_fp32 else round8(c.out_channels)
        groups = cin if self.depthwise else c.groups
        return cin, cout, groups

    def _sources(self):
        ts = [self.conv.weight, self.conv.bias]
        if self.bn is not None:
            ts += [self.bn.weight, self.bn.bias, self.bn.running_mean, self.bn.running_var]
        return ts

    def _state_key(self, dtype, cpitch, d: ConvDesc):
        # everything the packed blob and the scale/shift arrays depend on besides the source tensors: physical channel counts
        # (the fp32 classifier output keeps its logical width), output type, and the parity of the left padding (the stem
        # packing depends on it; TF-"same" mode changes it per call)
        return (dtype, cpitch, d.Cin, d.Cout, d.groups, d.out_dtype, d.pad_l & 1) + \
            tuple((t.data_ptr(), t._version) if t is not None else None for t in self._sources())

    def desc(self, x: NHWC, act, post_act, has_res, out_code=None, logits=False, pad4=None) -> ConvDesc:
        """`pad4`: per-call (left, right, top, bottom) padding overriding the block's own (never stored on the runner)."""
        c = self.conv
        kh, kw = _pair(c.kernel_size)
        sh, sw = _pair(c.stride)
        dh, dw = _pair(c.dilation)
        pad4 = pad4 if pad4 is not None else self.pad4
        if pad4 is not None:                    # ConvBlock's 4-tuple padding: (left, right, top, bottom)
            pl, pr, pt, pb = [int(v) for v in pad4]
        else:
            ph, pw = _pair(c.padding)
            pt = pb = ph
            pl = pr = pw
        code = _CODE_OF_TORCH[x.dtype]
        cin, cout, groups = self._phys(x.cpitch, logits)          # `logits`: the fp32 classifier output keeps its logical width
        if x.cpitch != cin and not (x.cpitch == 4 and cin <= 4):
            raise RuntimeError("input handle has {} physical channels, this convolution expects {}".format(x.cpitch, cin))
        d = ConvDesc(N=x.N, H=x.H, W=x.W, Cin=cin, Cout=cout, kh=kh, kw=kw, stride_h=sh, stride_w=sw,
                     pad_t=pt, pad_l=pl, pad_b=pb, pad_r=pr, dil_h=dh, dil_w=dw, groups=groups, act=act, post_act=post_act,
                     has_residual=1 if has_res else 0, dtype=code, out_dtype=code if out_code is None else out_code,
                     x_cpitch=x.cpitch, x_wpitch=x.wpitch)
        return d

    def prepare(self, x: NHWC, d: ConvDesc):
        if isinstance(c_mode := self.conv.padding_mode, str) and c_mode != "zeros":
            raise NotImplementedError("padding_mode {}".format(c_mode))
        key = self._state_key(x.dtype, x.cpitch, d)
        if key == self._key:
            return
        if self._foreign:
            # the packed state came from another rank; a new key (another dtype / channel pitch / padding parity / output type, or
            # touched parameters) would re-pack from THIS rank's fp32 tensors, which broadcast_packed_state never updated: wrong
            # logits with no error. Refuse; the caller re-synchronises the fp32 state first.
            raise RuntimeError(
                "this convolution's packed weights were received by parallel.broadcast_packed_state and would have to be re-packed "
                "(the input's dtype / channel pitch / padding parity or a parameter changed), but this rank's fp32 parameters are "
                "not the source's. Call parallel.broadcast_module_state(net) (or load_state_dict on every rank) before running "
                "another configuration.")
        dev = x.device
        w = self.conv.weight
        if w.device != dev:
            raise RuntimeError("model parameters are on {} but the input is on {}".format(w.device, dev))
        L, ctx, st = _lib.lib(), _ctx(dev), _stream(dev)
        nbytes = ctypes.c_size_t(0)
        fn_bytes = L.pcv_dwconv_packed_bytes if self.depthwise else L.pcv_conv_packed_bytes
        if fn_bytes(ctypes.byref(d), ctypes.byref(nbytes)) != 0:
            raise _lib.PcvError(-1, "unsupported convolution configuration: {}".format(
                {f[0]: getattr(d, f[0]) for f in d._fields_}))
        packed = torch.empty((nbytes.value + 15) // 16 * 16, dtype=torch.uint8, device=dev)
        w32 = w.detach().float()
        if w32.dim() == 2:                                          # nn.Linear viewed as a 1x1 convolution
            w32 = w32[:, :, None, None]
        C, Cl = d.Cout, self.conv.out_channels                      # physical / logical output channels
        cin_g = d.Cin // d.groups                                   # physical input channels per group
        if w32.shape[0] != C or w32.shape[1] != cin_g:              # zero rows / columns for the pad channels
            w32 = F.pad(w32, (0, 0, 0, 0, 0, cin_g - w32.shape[1], 0, C - w32.shape[0]))
        w32 = w32.contiguous()
        fn_pack = L.pcv_dwconv_pack if self.depthwise else L.pcv_conv_pack
        _lib.check(fn_pack(ctx, ctypes.byref(d), _ptr(w32), _ptr(packed), st), ctx)

        def padded(t, fill):
            t = t.detach().float()
            return (F.pad(t, (0, C - Cl), value=fill) if C != Cl else t).contiguous()
        scale = torch.empty(C, dtype=torch.float32, device=dev)
        shift = torch.empty(C, dtype=torch.float32, device=dev)
        bias = padded(self.conv.bias, 0.0) if self.conv.bias is not None else None
        if self.bn is not None:
            if not isinstance(self.bn, nn.BatchNorm2d):
                raise NotImplementedError("only BatchNorm2d folds into the conv epilogue, got {}".format(type(self.bn).__name__))
            # pad channels: gamma 0, beta 0 -> scale 0, shift 0
            g = padded(self.bn.weight, 0.0) if self.bn.weight is not None else padded(torch.ones(Cl, device=dev), 0.0)
            b = padded(self.bn.bias, 0.0) if self.bn.bias is not None else torch.zeros(C, device=dev)
            m = padded(self.bn.running_mean, 0.0)
            v = padded(self.bn.running_var, 1.0)
            _lib.check(L.pcv_bn_fold(ctx, C, _ptr(g), _ptr(b), _ptr(m), _ptr(v), ctypes.c_float(self.bn.eps), _ptr(bias),
                                     _ptr(scale), _ptr(shift), st), ctx)
        else:
            _lib.check(L.pcv_bn_fold(ctx, C, None, None, None, None, ctypes.c_float(0.0), _ptr(bias), _ptr(scale),
                                     _ptr(shift), st), ctx)
        torch.cuda.current_stream(dev).synchronize()      # w32 & friends are temporaries; load-time only
        self.packed, self.scale, self.shift, self._key = packed, scale, shift, key

    def adopt_foreign_state(self):
        """Called after packed / scale / shift were overwritten with another rank's (parallel.broadcast_packed_state): the
        cache key stays valid for exactly the configuration it was built for, anything else raises in prepare()."""
        self._foreign = True
        self._sq_key = None

    def local_state_restored(self):
        """The fp32 sources are authoritative again (parallel.broadcast_module_state, load_state_dict): re-pack on next use."""
        self._foreign = False
        self._key = None

    def run(self, x: NHWC, act=0, residual: NHWC | None = None, post_act=0, out_fp32=False, pad4=None, out=None, gate=None) -> NHWC:
        """`pad4`: explicit (left, right, top, bottom) zero padding for this call (the `F.pad` a unit applies in front of
        a padding-0 convolution, efficientnet.py:108-109,189-190,236-237); it stays inside the kernel's bounds checks."""
        if self.bn is not None and self.bn.training:
            raise RuntimeError("pytorchcv_amd is an inference path: call net.eval() first (BatchNorm is folded)")
        if isinstance(x, LazyNCHW) and not x.materialized and residual is None and out is None and gate is None and not out_fp32 \
                and post_act == 0 and pad4 is None:
            y = self._stem_from_nchw(x, act, pool=False)
            if y is not None:
                return y
        d = self.desc(x, act, post_act, residual is not None, out_code=0 if out_fp32 else None, logits=out_fp32, pad4=pad4)
        self.prepare(x, d)
        return self._launch(x, d, residual, out, gate)

    def _stem_from_nchw(self, x: "LazyNCHW", act: int, pool: bool):
        """The stem convolution (+ the init block's MaxPool2d(3, 2, 1) with `pool`) straight from the fp32 NCHW image
        (pcv_conv2d_nchw_stem_fused); None when this convolution is not the covered stem shape."""
        if self.depthwise or self.pad4 is not None or x.src.numel() * 4 >= (1 << 31):      # (one launch addresses < 2 GiB: the
            return None                                                                      # converted path splits the batch)
        d = self.desc(x, act, 0, False)
        L, ctx = _lib.lib(), _ctx(x.device)
        if not L.pcv_conv2d_nchw_stem_supported(ctypes.byref(d), 1 if pool else 0):
            return None
        self.prepare(x, d)
        Ho = (x.H + d.pad_t + d.pad_b - d.dil_h * (d.kh - 1) - 1) // d.stride_h + 1
        Wo = (x.W + d.pad_l + d.pad_r - d.dil_w * (d.kw - 1) - 1) // d.stride_w + 1
        if pool:
            Ho, Wo = (Ho + 2 - 3) // 2 + 1, (Wo + 2 - 3) // 2 + 1
        if Ho <= 0 or Wo <= 0:
            return None
        y = torch.empty((x.N, Ho, Wo, d.Cout), dtype=x.dtype, device=x.device)
        _lib.check(L.pcv_conv2d_nchw_stem_fused(ctx, ctypes.byref(d), _ptr(x.src), _ptr(self.packed), _ptr(self.scale), _ptr(self.shift),
                                                _ptr(y), 1 if pool else 0, _stream(x.device)), ctx)
        return NHWC(y, x.N, Ho, Wo, self.conv.out_channels, cpitch=d.Cout)

    def run_maxpool(self, x: NHWC, act: int, k: int, s: int, p: int, ceil_mode: bool = False):
        """This convolution + BN + activation and the MaxPool2d(k, s, p) behind it as ONE launch when covered
        (pcv_conv2d_maxpool_fused: the stem convolution with MaxPool2d(3, 2, 1)); returns the pooled handle or None."""
        if not FUSE_UNITS or self.depthwise or self.pad4 is not None:
            return None
        if self.bn is not None and self.bn.training:
            raise RuntimeError("pytorchcv_amd is an inference path: call net.eval() first (BatchNorm is folded)")
        if isinstance(x, LazyNCHW) and not x.materialized and (k, s, p) == (3, 2, 1) and not ceil_mode:
            y = self._stem_from_nchw(x, act, pool=True)
            if y is not None:
                return y
        d = self.desc(x, act, 0, False)
        L, ctx = _lib.lib(), _ctx(x.device)
        if not L.pcv_conv2d_maxpool_supported(ctypes.byref(d), k, s, p, 1 if ceil_mode else 0):
            return None
        self.prepare(x, d)
        Ho = (x.H + d.pad_t + d.pad_b - d.dil_h * (d.kh - 1) - 1) // d.stride_h + 1
        Wo = (x.W + d.pad_l + d.pad_r - d.dil_w * (d.kw - 1) - 1) // d.stride_w + 1
        Hq, Wq = (Ho + 2 * p - k) // s + 1, (Wo + 2 * p - k) // s + 1
        if Hq <= 0 or Wq <= 0:
            return None
        y = torch.empty((x.N, Hq, Wq, d.Cout), dtype=x.dtype, device=x.device)
        _lib.check(L.pcv_conv2d_maxpool_fused(ctx, ctypes.byref(d), _ptr(x.t), _ptr(self.packed), _ptr(self.scale), _ptr(self.shift),
                                              _ptr(y), k, s, p, 1 if ceil_mode else 0, _stream(x.device)), ctx)
        return NHWC(y, x.N, Hq, Wq, self.conv.out_channels, cpitch=d.Cout)

    def run_pair(self, x: NHWC, residual: NHWC, act: int, post_act: int, nxt: "ConvRunner", nxt_act: int, gate=None):
        """This convolution (+ residual, + post_act) and the 1x1 convolution `nxt` that consumes its output, as ONE launch
        (pcv_conv1x1_pair_fused): returns (y1, y2), or None when the pair of shapes is not covered by the fused kernel."""
        if not FUSE_UNITS or residual is None or self.depthwise or nxt.depthwise or self.pad4 is not None or nxt.pad4 is not None:
            return None
        if (self.bn is not None and self.bn.training) or (nxt.bn is not None and nxt.bn.training):
            raise RuntimeError("pytorchcv_amd is an inference path: call net.eval() first (BatchNorm is folded)")
        c = self.conv
        L, ctx, st = _lib.lib(), _ctx(x.device), _stream(x.device)
        d1 = self.desc(x, act, post_act, True)
        d2 = nxt.desc(_ShapeOnly(x.N, x.H, x.W, c.out_channels, x.dtype), nxt_act, 0, False)
        supported = L.pcv_conv1x1_pair_gated_supported if gate is not None else L.pcv_conv1x1_pair_supported
        if not supported(ctypes.byref(d1), ctypes.byref(d2)):
            return None
        if not residual.dense or residual.dtype != x.dtype or tuple(residual.t.shape) != (x.N, x.H, x.W, d1.Cout):
            raise RuntimeError("residual shape/dtype mismatch")
        if gate is not None and (gate.dtype != torch.float32 or tuple(gate.shape) != (x.N, d1.Cout) or not gate.is_contiguous()):
            raise RuntimeError("gate must be a contiguous fp32 [N, {}] tensor".format(d1.Cout))
        t1 = torch.empty((x.N, x.H, x.W, d1.Cout), dtype=x.dtype, device=x.device)
        t2 = torch.empty((x.N, x.H, x.W, d2.Cout), dtype=x.dtype, device=x.device)
        y1 = NHWC(t1, x.N, x.H, x.W, c.out_channels, cpitch=d1.Cout)
        self.prepare(x, d1)
        nxt.prepare(y1, d2)
        if gate is not None:
            _lib.check(L.pcv_conv1x1_pair_gated_fused(ctx, ctypes.byref(d1), ctypes.byref(d2), _ptr(x.t), _ptr(self.packed),
                                                      _ptr(self.scale), _ptr(self.shift), _ptr(gate), _ptr(residual.t), _ptr(t1),
                                                      _ptr(nxt.packed), _ptr(nxt.scale), _ptr(nxt.shift), _ptr(t2), st), ctx)
        else:
            _lib.check(L.pcv_conv1x1_pair_fused(ctx, ctypes.byref(d1), ctypes.byref(d2), _ptr(x.t), _ptr(self.packed),
                                                _ptr(self.scale), _ptr(self.shift), _ptr(residual.t), _ptr(t1), _ptr(nxt.packed),
                                                _ptr(nxt.scale), _ptr(nxt.shift), _ptr(t2), st), ctx)
        return y1, NHWC(t2, x.N, x.H, x.W, nxt.conv.out_channels, cpitch=d2.Cout)

    def run_pair_idconv(self, x: NHWC, x0: NHWC, idr: "ConvRunner", act: int, post_act: int, nxt: "ConvRunner", nxt_act: int):
        """`run_pair` for the first unit of a stage: the skip tensor is `idr` (1x1 convolution + BN, no activation) applied to
        the unit's input `x0`, recomputed inside the fused kernel instead of being written and read back
        (pcv_conv1x1_pair_idconv_fused). Returns (y1, y2), or None when the shapes are not covered."""
        if not FUSE_UNITS or any(r.depthwise or r.pad4 is not None for r in (self, idr, nxt)):
            return None
        if any(r.bn is not None and r.bn.training for r in (self, idr, nxt)):
            raise RuntimeError("pytorchcv_amd is an inference path: call net.eval() first (BatchNorm is folded)")
        if not x0.dense or x0.dtype != x.dtype or (x0.N, x0.H, x0.W) != (x.N, x.H, x.W):
            return None
        c = self.conv
        L, ctx, st = _lib.lib(), _ctx(x.device), _stream(x.device)
        di = idr.desc(x0, 0, 0, False)
        d1 = self.desc(x, act, post_act, True)
        d2 = nxt.desc(_ShapeOnly(x.N, x.H, x.W, c.out_channels, x.dtype), nxt_act, 0, False)
        if not L.pcv_conv1x1_pair_idconv_supported(ctypes.byref(di), ctypes.byref(d1), ctypes.byref(d2)):
            return None
        t1 = torch.empty((x.N, x.H, x.W, d1.Cout), dtype=x.dtype, device=x.device)
        t2 = torch.empty((x.N, x.H, x.W, d2.Cout), dtype=x.dtype, device=x.device)
        y1 = NHWC(t1, x.N, x.H, x.W, c.out_channels, cpitch=d1.Cout)
        idr.prepare(x0, di)
        self.prepare(x, d1)
        nxt.prepare(y1, d2)
        _lib.check(L.pcv_conv1x1_pair_idconv_fused(ctx, ctypes.byref(di), ctypes.byref(d1), ctypes.byref(d2), _ptr(x0.t),
                                                   _ptr(idr.packed), _ptr(idr.scale), _ptr(idr.shift), _ptr(x.t), _ptr(self.packed),
                                                   _ptr(self.scale), _ptr(self.shift), _ptr(t1), _ptr(nxt.packed), _ptr(nxt.scale),
                                                   _ptr(nxt.shift), _ptr(t2), st), ctx)
        return y1, NHWC(t2, x.N, x.H, x.W, nxt.conv.out_channels, cpitch=d2.Cout)

    def _launch(self, x: NHWC, d: ConvDesc, residual, out=None, gate=None):
        """`out` = (tensor [N, Ho, Wo, Ctot], channel offset): write the result into that channel slice of a wider
        (concatenation) buffer instead of a fresh tensor - `torch.cat((identity, x), dim=1)` without the copy."""
        c = self.conv
        kh, kw = d.kh, d.kw
        Ho = (x.H + d.pad_t + d.pad_b - d.dil_h * (kh - 1) - 1) // d.stride_h + 1
        Wo = (x.W + d.pad_l + d.pad_r - d.dil_w * (kw - 1) - 1) // d.stride_w + 1
        if Ho <= 0 or Wo <= 0:
            raise RuntimeError("convolution output would be empty")
        out_dt = torch.float32 if d.out_dtype == 0 else x.dtype
        if out is not None:
            buf, coff = out
            if self.depthwise or tuple(buf.shape[:3]) != (x.N, Ho, Wo) or buf.dtype != out_dt or not buf.is_contiguous() or \
                    coff % 8 != 0 or c.out_channels % 8 != 0 or coff + c.out_channels > buf.shape[3] or buf.device != x.device:
                raise RuntimeError("bad concatenation slice for a convolution output (channel counts must be multiples of 8)")
            d.y_cpitch = int(buf.shape[3])
            y = buf[:, :, :, coff:]                 # a view: its data_ptr is the first element of the slice
        else:
            y = torch.empty((x.N, Ho, Wo, d.Cout), dtype=out_dt, device=x.device)          # d.Cout: physical channels
        if residual is not None:
            if not residual.dense or tuple(residual.t.shape) != (x.N, Ho, Wo, d.Cout) or residual.dtype != x.dtype:
                raise RuntimeError("residual shape/dtype mismatch: {} vs {}".format(
                    tuple(residual.t.shape), (x.N, Ho, Wo, d.Cout)))
        if gate is not None:
            if self.depthwise or gate.dtype != torch.float32 or tuple(gate.shape) != (x.N, d.Cout) or not gate.is_contiguous():
                raise RuntimeError("gate must be a contiguous fp32 [N, {}] tensor of a dense convolution".format(d.Cout))
        self._launch_range(x, d, residual, y, 0, x.N, gate)
        return None if out is not None else NHWC(y, x.N, Ho, Wo, c.out_channels, cpitch=d.Cout)

    def _launch_range(self, x, d, residual, y, n0, n1, gate=None):
        """Launch images [n0, n1); halve the range when one launch would exceed the 2 GiB addressing window."""
        L, ctx, st = _lib.lib(), _ctx(x.device), _stream(x.device)
        d.N = n1 - n0
        rp = _ptr(residual.t[n0:n1]) if residual is not None else None
        if gate is not None:
            rc = L.pcv_conv2d_gated_fused(ctx, ctypes.byref(d), _ptr(x.t[n0:n1]), _ptr(self.packed), _ptr(self.scale),
                                          _ptr(self.shift), _ptr(gate[n0:n1]), rp, _ptr(y[n0:n1]), st)
        else:
            fn = L.pcv_dwconv2d_fused if self.depthwise else L.pcv_conv2d_fused
            rc = fn(ctx, ctypes.byref(d), _ptr(x.t[n0:n1]), _ptr(self.packed), _ptr(self.scale), _ptr(self.shift), rp, _ptr(y[n0:n1]), st)
        if rc == _lib.PCV_ERR_TOO_LARGE and n1 - n0 > 1:
            mid = (n0 + n1) // 2
            self._launch_range(x, d, residual, y, n0, mid, gate)
            self._launch_range(x, d, residual, y, mid, n1, gate)
            return
        _lib.check(rc, ctx)

    def squeezed_excite(self, z: NHWC, w1, b1, w2, b2, mid_act: int, out_act: int) -> torch.Tensor:
        """SE gate of `SEBlock(BN(conv(z)))` for a 1x1 stride-1 convolution WITHOUT activation, computed before the convolution
        runs: mean_hw(BN(conv(z))) = BN(conv(mean_hw(z))) (both affine), and the first excitation layer absorbs that map:
        mid = mid_act(W1 . (S W mean + shift) + b1) = mid_act((W1 S W) . mean + (W1 shift + b1)) with the [M, Cin] product
        folded once at load time. Two small fp32 layers on the squeezed INPUT: gate fp32 [N, Cout_physical]."""
        c = self.conv
        if self.depthwise or c.groups != 1 or tuple(_pair(c.kernel_size)) != (1, 1) or tuple(_pair(c.stride)) != (1, 1) or \
                self.pad4 is not None or tuple(_pair(c.padding)) != (0, 0):
            raise RuntimeError("squeezed_excite needs a plain 1x1 stride-1 convolution")
        d = self.desc(z, 0, 0, False)
        self.prepare(z, d)                                   # scale / shift of the folded BatchNorm (+ bias)
        key = ("sq", self._key) + tuple((t.data_ptr(), t._version) for t in (w1, w2, b1, b2))
        if getattr(self, "_sq_key", None) != key:
            Cl, CP = c.out_channels, d.Cout
            w = c.weight.detach().float().reshape(Cl, -1)
            w = F.pad(w, (0, d.Cin - w.shape[1], 0, CP - Cl)) * self.scale[:, None]       # [CP, Cin]: BN scale folded into the rows
            w1p = F.pad(w1.float(), (0, CP - Cl))                                          # [M, CP]
            self._sq_w1 = (w1p @ w).contiguous()                                           # [M, Cin]
            self._sq_b1 = (w1p @ self.shift + b1.float()).contiguous()
            self._sq_w2 = F.pad(w2.float(), (0, 0, 0, CP - Cl)).contiguous()               # [CP, M]
            self._sq_b2 = F.pad(b2.float(), (0, CP - Cl)).contiguous()
            self._sq_key = key
        L, ctx, st = _lib.lib(), _ctx(z.device), _stream(z.device)
        M = self._sq_w1.shape[0]
        mean = torch.empty((z.N, d.Cin), dtype=torch.float32, device=z.device)
        _lib.check(L.pcv_se_squeeze(ctx, _ptr(z.t), _ptr(mean), z.N, z.H * z.W, d.Cin, _CODE_OF_TORCH[z.dtype], st), ctx)
        mid = torch.empty((z.N, M), dtype=torch.float32, device=z.device)
        gate = torch.empty((z.N, d.Cout), dtype=torch.float32, device=z.device)
        _lib.check(L.pcv_fc_f32(ctx, _ptr(mean), _ptr(self._sq_w1), _ptr(self._sq_b1), _ptr(mid), z.N, d.Cin, M, mid_act, st), ctx)
        _lib.check(L.pcv_fc_f32(ctx, _ptr(mid), _ptr(self._sq_w2), _ptr(self._sq_b2), _ptr(gate), z.N, M, d.Cout, out_act, st), ctx)
        return gate


def _pool_out(n: int, k: int, s: int, p: int, ceil_mode: bool) -> int:
    o = (n + 2 * p - k + (s - 1 if ceil_mode else 0)) // s + 1
    if ceil_mode and (o - 1) * s >= n + p:
        o -= 1
    return o


def channel_slice(x: NHWC, offset: int, count: int) -> NHWC:
    """x[:, offset:offset+count] (channel dimension) as a canonical handle of its own (torch.chunk / split)."""
    if x.wpitch != x.W or offset < 0 or offset + count > x.C:
        raise RuntimeError("bad channel slice")
    cp = round8(count)
    y = torch.empty((x.N, x.H, x.W, cp), dtype=x.dtype, device=x.device)
    ctx = _ctx(x.device)
    _lib.check(_lib.lib().pcv_channel_slice(ctx, _ptr(x.t), _ptr(y), x.N * x.H * x.W, count, offset, x.cpitch, cp,
                                            _CODE_OF_TORCH[x.dtype], _stream(x.device)), ctx)
    return NHWC(y, x.N, x.H, x.W, count, cpitch=cp)


def channel_concat_into(x: NHWC, buf: torch.Tensor, coff: int):
    """buf[:, :, :, coff:coff + x.C] = x (the copy form of torch.cat along channels; a convolution writes its slice itself
    through `ConvRunner.run(out=(buf, coff))`). x.C and coff must be multiples of 8."""
    if x.wpitch != x.W or x.C % 8 or coff % 8 or tuple(buf.shape[:3]) != (x.N, x.H, x.W) or buf.dtype != x.dtype or \
            coff + x.C > buf.shape[3] or not buf.is_contiguous():
        raise RuntimeError("channel concatenation needs channel counts / offsets that are multiples of 8 and equal maps")
    ctx = _ctx(x.device)
    _lib.check(_lib.lib().pcv_channel_concat(ctx, _ptr(x.t), _ptr(buf), x.N * x.H * x.W, x.C, x.cpitch, int(buf.shape[3]), coff,
                                             _CODE_OF_TORCH[x.dtype], _stream(x.device)), ctx)


def interpolate(x: NHWC, out_size, bilinear: bool, align_corners: bool) -> NHWC:
    """F.interpolate(size=out_size, mode="bilinear" | "nearest", align_corners) on an NHWC handle (pcv_interpolate)."""
    if not x.dense:
        raise RuntimeError("interpolation on a padded handle")
    Ho, Wo = int(out_size[0]), int(out_size[1])
    y = torch.empty((x.N, Ho, Wo, x.cpitch), dtype=x.dtype, device=x.device)
    ctx = _ctx(x.device)
    _lib.check(_lib.lib().pcv_interpolate(ctx, _ptr(x.t), _ptr(y), x.N, x.H, x.W, x.cpitch, Ho, Wo, 1 if bilinear else 0,
                                          1 if align_corners else 0, _CODE_OF_TORCH[x.dtype], _stream(x.device)), ctx)
    return NHWC(y, x.N, Ho, Wo, x.C, cpitch=x.cpitch)


def add(a: NHWC, b: NHWC, post_act: int = 0) -> NHWC:
    """post_act(a + b) as one pass (pcv_se_scale with a unit gate): the fallback of a summing `Concurrent` whose branch does not
    end in a convolution (a convolution takes the running sum as its epilogue residual instead)."""
    if a.t.shape != b.t.shape or a.dtype != b.dtype or not a.dense:
        raise RuntimeError("add: operands do not match")
    ones = torch.ones((a.N, a.cpitch), dtype=torch.float32, device=a.device)
    y = torch.empty_like(a.t)
    ctx = _ctx(a.device)
    _lib.check(_lib.lib().pcv_se_scale(ctx, _ptr(a.t), _ptr(ones), _ptr(b.t), _ptr(y), a.N, a.H * a.W, a.cpitch, post_act,
                                       _CODE_OF_TORCH[a.dtype], _stream(a.device)), ctx)
    return NHWC(y, a.N, a.H, a.W, a.C, cpitch=a.cpitch)


def cat_shuffle2(a: NHWC, b: NHWC, half: int) -> NHWC:
    """channel_shuffle(torch.cat((a[:, :half], b[:, :half]), dim=1), groups=2) in one pass."""
    if (a.N, a.H, a.W) != (b.N, b.H, b.W) or a.dtype != b.dtype or a.wpitch != a.W or b.wpitch != b.W or half > a.C or half > b.C:
        raise RuntimeError("cat/shuffle operands do not match")
    cp = round8(2 * half)
    y = torch.empty((a.N, a.H, a.W, cp), dtype=a.dtype, device=a.device)
    ctx = _ctx(a.device)
    _lib.check(_lib.lib().pcv_channel_interleave2(ctx, _ptr(a.t), _ptr(b.t), _ptr(y), a.N * a.H * a.W, half, a.cpitch, b.cpitch,
                                                  cp, _CODE_OF_TORCH[a.dtype], _stream(a.device)), ctx)
    return NHWC(y, a.N, a.H, a.W, 2 * half, cpitch=cp)


def maxpool2d(x: NHWC, k: int, s: int, p: int, ceil_mode: bool = False) -> NHWC:
    if not x.dense:
        raise RuntimeError("max-pool on a padded handle")
    Ho, Wo = _pool_out(x.H, k, s, p, ceil_mode), _pool_out(x.W, k, s, p, ceil_mode)
    y = torch.empty((x.N, Ho, Wo, x.cpitch), dtype=x.dtype, device=x.device)
    ctx = _ctx(x.device)
    _lib.check(_lib.lib().pcv_maxpool2d(ctx, _ptr(x.t), _ptr(y), x.N, x.H, x.W, x.cpitch, k, s, p, 1 if ceil_mode else 0,
                                        _CODE_OF_TORCH[x.dtype], _stream(x.device)), ctx)
    return NHWC(y, x.N, Ho, Wo, x.C, cpitch=x.cpitch)


def avgpool2d(x: NHWC, k: int, s: int, out_fp32: bool = False) -> NHWC:
    """`out_fp32`: write the pooled map in fp32 (the classifier input: see FP32_HEAD)."""
    if not x.dense:
        raise RuntimeError("avg-pool on a padded handle")
    if k > x.H or k > x.W:
        raise RuntimeError("AvgPool2d kernel {} larger than the {}x{} map".format(k, x.H, x.W))
    Ho, Wo = (x.H - k) // s + 1, (x.W - k) // s + 1
    y = torch.empty((x.N, Ho, Wo, x.cpitch), dtype=torch.float32 if out_fp32 else x.dtype, device=x.device)
    ctx = _ctx(x.device)
    code = _CODE_OF_TORCH[x.dtype]
    _lib.check(_lib.lib().pcv_avgpool2d(ctx, _ptr(x.t), _ptr(y), x.N, x.H, x.W, x.cpitch, k, s, code, 0 if out_fp32 else code,
                                        _stream(x.device)), ctx)
    return NHWC(y, x.N, Ho, Wo, x.C, cpitch=x.cpitch)


# The classifier head runs in fp32: the `final_pool` of a net writes fp32 pooled features and the classifier behind it (Linear /
# 1x1 convolutions on the 1 x 1 map: 2-3 MMAC per image, nothing) runs on the exact-f32 MFMA path with fp32 weights. Measured on the
# golden fixtures (tests/tools/bf16_drift.py): everything in front of the global pool is averaged over the 49 positions of the last
# map, the rounding of the pooled vector and of the classifier weights is not - those two roundings alone were 43 % of the
# noise power of ResNet-50's bf16 logits (max |d| vs the fp32 reference 1.1e-2 -> 8.1e-3, ResNeXt-101 1.3e-2 -> 8.9e-3).
FP32_HEAD = os.environ.get("PCV_AMD_FP32_HEAD", "1") != "0"
STEM_FROM_NCHW = os.environ.get("PCV_AMD_STEM_NCHW", "1") != "0"     # the stem reads the fp32 NCHW input itself (0: layout kernel first)


# Unit-level fusions (pcv_mbconv_fused, pcv_conv1x1_pair_fused) can be switched off to time the per-layer kernels on their own
# (bench.py's roofline pass over the depthwise kernel class does that); the results are the same either way.
FUSE_UNITS = os.environ.get("PCV_AMD_FUSE_UNITS", "1") != "0"


def mbconv_fused(exp, exp_act: int, dw, dw_act: int, proj, proj_act: int, x: NHWC, residual, post_act: int):
    """[expand 1x1 ->] depthwise 3x3 -> project 1x1 (+ residual) as ONE launch (pcv_mbconv_fused); `exp`, `dw`, `proj` are
    the ConvRunners of the three blocks (`exp` may be None). Returns the unit output, or None when the triple of shapes is
    not covered by the fused kernel (the caller then issues the separate launches)."""
    if not FUSE_UNITS or not x.dense or not dw.depthwise or proj.depthwise or (exp is not None and exp.depthwise):
        return None
    if any(r is not None and r.pad4 is not None for r in (exp, dw, proj)):
        return None
    for r in (exp, dw, proj):
        if r is not None and r.bn is not None and r.bn.training:
            raise RuntimeError("pytorchcv_amd is an inference path: call net.eval() first (BatchNorm is folded)")
    L, ctx, st = _lib.lib(), _ctx(x.device), _stream(x.device)
    cur = x
    d_exp = None
    if exp is not None:
        d_exp = exp.desc(x, exp_act, 0, False)
        cur = _ShapeOnly(x.N, x.H, x.W, exp.conv.out_channels, x.dtype)
    d_dw = dw.desc(cur, dw_act, 0, False)
    Ho = (cur.H + d_dw.pad_t + d_dw.pad_b - d_dw.dil_h * (d_dw.kh - 1) - 1) // d_dw.stride_h + 1
    Wo = (cur.W + d_dw.pad_l + d_dw.pad_r - d_dw.dil_w * (d_dw.kw - 1) - 1) // d_dw.stride_w + 1
    mid = _ShapeOnly(x.N, Ho, Wo, dw.conv.out_channels, x.dtype)
    d_proj = proj.desc(mid, proj_act, post_act, residual is not None)
    if not L.pcv_mbconv_supported(ctypes.byref(d_exp) if d_exp is not None else None, ctypes.byref(d_dw), ctypes.byref(d_proj)):
        return None
    Cout = d_proj.Cout                                           # physical
    if residual is not None and (not residual.dense or residual.dtype != x.dtype or
                                 tuple(residual.t.shape) != (x.N, Ho, Wo, Cout)):
        raise RuntimeError("residual shape/dtype mismatch")
    # weights / BN constants are packed through the runners' own caches (dense handles of the right shape and dtype)
    if exp is not None:
        exp.prepare(x, d_exp)
    dw.prepare(_PrepHandle(cur, x.device), d_dw)
    proj.prepare(_PrepHandle(mid, x.device), d_proj)
    y = torch.empty((x.N, Ho, Wo, Cout), dtype=x.dtype, device=x.device)
    _lib.check(L.pcv_mbconv_fused(ctx, ctypes.byref(d_exp) if d_exp is not None else None, ctypes.byref(d_dw), ctypes.byref(d_proj),
                                  _ptr(x.t), _ptr(exp.packed) if exp is not None else None,
                                  _ptr(exp.scale) if exp is not None else None, _ptr(exp.shift) if exp is not None else None,
                                  _ptr(dw.packed), _ptr(dw.scale), _ptr(dw.shift), _ptr(proj.packed), _ptr(proj.scale),
                                  _ptr(proj.shift), _ptr(residual.t) if residual is not None else None, _ptr(y), st), ctx)
    return NHWC(y, x.N, Ho, Wo, proj.conv.out_channels, cpitch=Cout)


class _PrepHandle(object):
    """dtype / cpitch / device of a tensor that is never materialised, for ConvRunner.prepare."""
    __slots__ = ("dtype", "cpitch", "device")

    def __init__(self, shape_only, device):
        self.dtype, self.cpitch, self.device = shape_only.dtype, shape_only.cpitch, device


class BnActRunner(object):
    """Eval-mode BatchNorm2d + activation as one elementwise launch (pcv_bn_act); scale/shift are folded once and refolded
    when the parameters change (same cache rule as ConvRunner)."""
    def __init__(self, bn):
        self.bn = bn
        self._key = None
        self._foreign = False       # see ConvRunner._foreign
        self.scale = self.shift = None

    def _state_key(self):
        ts = (self.bn.weight, self.bn.bias, self.bn.running_mean, self.bn.running_var)
        return tuple((t.data_ptr(), t._version) if t is not None else None for t in ts)

    def prepare(self, x: NHWC):
        key = self._state_key()
        if key == self._key:
            return
        if self._foreign:
            raise RuntimeError("this BatchNorm's folded constants were received by parallel.broadcast_packed_state and its local "
                               "parameters changed: call parallel.broadcast_module_state(net) first")
        bn, dev = self.bn, x.device
        if not isinstance(bn, nn.BatchNorm2d):
            raise NotImplementedError("only BatchNorm2d folds to scale/shift, got {}".format(type(bn).__name__))
        if bn.running_mean.device != dev:
            raise RuntimeError("model parameters are on {} but the input is on {}".format(bn.running_mean.device, dev))
        Cl = bn.num_features
        C = round8(Cl)                                          # pad channels: scale 0, shift 0

        def padded(t, fill):
            t = t.detach().float()
            return (F.pad(t, (0, C - Cl), value=fill) if C != Cl else t).contiguous()
        L, ctx, st = _lib.lib(), _ctx(dev), _stream(dev)
        g = padded(bn.weight if bn.weight is not None else torch.ones(Cl, device=dev), 0.0)
        b = padded(bn.bias if bn.bias is not None else torch.zeros(Cl, device=dev), 0.0)
        m = padded(bn.running_mean, 0.0)
        v = padded(bn.running_var, 1.0)
        scale = torch.empty(C, dtype=torch.float32, device=dev)
        shift = torch.empty(C, dtype=torch.float32, device=dev)
        _lib.check(L.pcv_bn_fold(ctx, C, _ptr(g), _ptr(b), _ptr(m), _ptr(v), ctypes.c_float(bn.eps), None, _ptr(scale),
                                 _ptr(shift), st), ctx)
        torch.cuda.current_stream(dev).synchronize()
        self.scale, self.shift, self._key = scale, shift, key

    def adopt_foreign_state(self):
        self._foreign = True

    def local_state_restored(self):
        self._foreign = False
        self._key = None

    def run(self, x: NHWC, act: int) -> NHWC:
        if self.bn.training:
            raise RuntimeError("pytorchcv_amd is an inference path: call net.eval() first (BatchNorm is folded)")
        if x.wpitch != x.W:
            raise RuntimeError("BatchNorm + activation on a row-padded handle")
        if x.C != self.bn.num_features:
            raise RuntimeError("BatchNorm2d expects {} channels, got {}".format(self.bn.num_features, x.C))
        self.prepare(x)
        CP = round8(x.C)                        # physical channels processed (x may also be the prefix of a wider concat buffer)
        y = torch.empty((x.N, x.H, x.W, CP), dtype=x.dtype, device=x.device)
        ctx = _ctx(x.device)
        _lib.check(_lib.lib().pcv_bn_act(ctx, _ptr(x.t), _ptr(self.scale), _ptr(self.shift), _ptr(y), x.N * x.H * x.W, CP,
                                         x.cpitch, act, _CODE_OF_TORCH[x.dtype], _stream(x.device)), ctx)
        return NHWC(y, x.N, x.H, x.W, x.C, cpitch=CP)


def global_avgpool(x: NHWC, out_fp32: bool = False) -> NHWC:
    """nn.AdaptiveAvgPool2d(1) -> [N,1,1,C]. `out_fp32`: see avgpool2d."""
    if not x.dense:
        raise RuntimeError("avg-pool on a padded handle")
    y = torch.empty((x.N, 1, 1, x.cpitch), dtype=torch.float32 if out_fp32 else x.dtype, device=x.device)
    ctx = _ctx(x.device)
    code = _CODE_OF_TORCH[x.dtype]
    _lib.check(_lib.lib().pcv_global_avgpool(ctx, _ptr(x.t), _ptr(y), x.N, x.H * x.W, x.cpitch, code, 0 if out_fp32 else code,
                                             _stream(x.device)), ctx)
    return NHWC(y, x.N, 1, 1, x.C, cpitch=x.cpitch)


def se_forward(x: NHWC, w1, b1, w2, b2, mid_act: int, out_act: int, residual: NHWC | None, post_act: int) -> NHWC:
    """SEBlock on the hot path: squeeze -> excite (fp32) -> scale (+ residual, + activation)."""
    if not x.dense:
        raise RuntimeError("SE on a padded handle")
    L, ctx, st = _lib.lib(), _ctx(x.device), _stream(x.device)
    code = _CODE_OF_TORCH[x.dtype]
    CP = x.cpitch                                              # physical channels; the FC weights get zero columns / rows for the pads
    if CP != x.C:
        w1 = F.pad(w1, (0, CP - x.C)).contiguous()
        w2 = F.pad(w2, (0, 0, 0, CP - x.C)).contiguous()
        b2 = F.pad(b2, (0, CP - x.C)).contiguous()
    mean = torch.empty((x.N, CP), dtype=torch.float32, device=x.device)
    gate = torch.empty((x.N, CP), dtype=torch.float32, device=x.device)
    _lib.check(L.pcv_se_squeeze(ctx, _ptr(x.t), _ptr(mean), x.N, x.H * x.W, CP, code, st), ctx)
    M = w1.shape[0]
    mid = torch.empty((x.N, M), dtype=torch.float32, device=x.device)
    _lib.check(L.pcv_se_excite(ctx, _ptr(mean), _ptr(w1), _ptr(b1), _ptr(w2), _ptr(b2), _ptr(mid), _ptr(gate), x.N, CP, M,
                               mid_act, out_act, st), ctx)
    y = torch.empty_like(x.t)
    if residual is not None and (residual.t.shape != x.t.shape or residual.dtype != x.dtype):
        raise RuntimeError("SE residual shape/dtype mismatch")
    _lib.check(L.pcv_se_scale(ctx, _ptr(x.t), _ptr(gate), _ptr(residual.t) if residual is not None else None, _ptr(y),
                              x.N, x.H * x.W, CP, post_act, code, st), ctx)
    return NHWC(y, x.N, x.H, x.W, x.C, cpitch=CP)
