"""
    ctypes binding of the C ABI in include/pcv_amd.h (csrc/libpcv_amd.so, built in-tree by `__graft_entry__.build()` /
    `make -C pytorchcv_amd/csrc`). There is no CPU or eager fallback: if the library or a gfx950 device is missing,
    every entry point raises.
"""

__all__ = ['lib', 'ctx_for', 'check', 'ConvDesc', 'LIB_PATH', 'PcvError', 'PCV_ERR_TOO_LARGE']

import os
import ctypes
import threading

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libpcv_amd.so")
PCV_ERR_TOO_LARGE = -4

_c = ctypes
_VP = ctypes.c_void_p
_I = ctypes.c_int


class PcvError(RuntimeError):
    def __init__(self, code, msg):
        super(PcvError, self).__init__("pcv_amd error {}: {}".format(code, msg))
        self.code = code


PCV_ABI_VERSION = 4


class ConvDesc(ctypes.Structure):
    """Mirror of `pcv_conv_desc` (include/pcv_amd.h); `struct_size` is filled in by the constructor and checked by the
    library, and `lib()` compares sizeof(ConvDesc) with `pcv_conv_desc_size()` once at load time."""
    _fields_ = [(n, ctypes.c_int32) for n in (
        "struct_size", "N", "H", "W", "Cin", "Cout", "kh", "kw", "stride_h", "stride_w", "pad_t", "pad_l", "pad_b", "pad_r",
        "dil_h", "dil_w", "groups", "act", "post_act", "has_residual", "dtype", "out_dtype", "x_cpitch", "x_wpitch",
        "y_cpitch")]


    def __init__(self, *args, **kw):
        super(ConvDesc, self).__init__(*args, **kw)
        self.struct_size = ctypes.sizeof(ConvDesc)


_SIGS = {
    "pcv_abi_version": (_I, []),
    "pcv_conv_desc_size": (ctypes.c_size_t, []),
    "pcv_create": (_I, [ctypes.POINTER(_VP), _I]),
    "pcv_destroy": (_I, [_VP]),
    "pcv_last_error": (ctypes.c_char_p, [_VP]),
    "pcv_set_tuning": (_I, [_VP, ctypes.c_char_p, _I]),
    "pcv_fp16_guard_begin": (_I, [_VP, _VP, _VP]),
    "pcv_fp16_guard_end": (_I, [_VP, _VP, _VP, ctypes.c_long, _VP]),
    "pcv_fp16_overflow_count": (_I, [_VP, ctypes.POINTER(ctypes.c_uint), _VP]),
    "pcv_rccl_available": (_I, []),
    "pcv_rccl_broadcast": (_I, [_VP, _VP, ctypes.POINTER(_VP), ctypes.POINTER(ctypes.c_size_t), _I, _I, _VP]),
    "pcv_rccl_allgather": (_I, [_VP, _VP, _VP, _VP, ctypes.c_size_t, _VP]),
    "pcv_nchw_to_nhwc": (_I, [_VP, _VP, _VP, _I, _I, _I, _I, _I, _I, _I, _VP]),
    "pcv_nhwc_to_nchw": (_I, [_VP, _VP, _VP, _I, _I, _I, _I, _I, _I, _VP]),
    "pcv_preprocess_u8": (_I, [_VP, _VP, _VP, _I, _I, _I, _I, _I, _I, _I, _I, _I, _VP, _VP, _I, _VP]),
    "pcv_conv_packed_bytes": (_I, [ctypes.POINTER(ConvDesc), ctypes.POINTER(ctypes.c_size_t)]),
    "pcv_conv_pack": (_I, [_VP, ctypes.POINTER(ConvDesc), _VP, _VP, _VP]),
    "pcv_dwconv_packed_bytes": (_I, [ctypes.POINTER(ConvDesc), ctypes.POINTER(ctypes.c_size_t)]),
    "pcv_dwconv_pack": (_I, [_VP, ctypes.POINTER(ConvDesc), _VP, _VP, _VP]),
    "pcv_bn_fold": (_I, [_VP, _I, _VP, _VP, _VP, _VP, ctypes.c_float, _VP, _VP, _VP, _VP]),
    "pcv_conv2d_fused": (_I, [_VP, ctypes.POINTER(ConvDesc), _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "pcv_dwconv2d_fused": (_I, [_VP, ctypes.POINTER(ConvDesc), _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "pcv_maxpool2d": (_I, [_VP, _VP, _VP, _I, _I, _I, _I, _I, _I, _I, _I, _I, _VP]),
    "pcv_channel_slice": (_I, [_VP, _VP, _VP, ctypes.c_long, _I, _I, _I, _I, _I, _VP]),
    "pcv_channel_interleave2": (_I, [_VP, _VP, _VP, _VP, ctypes.c_long, _I, _I, _I, _I, _I, _VP]),
    "pcv_channel_concat": (_I, [_VP, _VP, _VP, ctypes.c_long, _I, _I, _I, _I, _I, _VP]),
    "pcv_interpolate": (_I, [_VP, _VP, _VP, _I, _I, _I, _I, _I, _I, _I, _I, _I, _VP]),
    "pcv_avgpool2d": (_I, [_VP, _VP, _VP, _I, _I, _I, _I, _I, _I, _I, _I, _VP]),
    "pcv_global_avgpool": (_I, [_VP, _VP, _VP, _I, _I, _I, _I, _I, _VP]),
    "pcv_gemm_bias": (_I, [_VP, _VP, _VP, _VP, _VP, _I, _I, _I, _I, _I, _VP]),
    "pcv_se_squeeze": (_I, [_VP, _VP, _VP, _I, _I, _I, _I, _VP]),
    "pcv_se_excite": (_I, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _I, _I, _I, _I, _I, _VP]),
    "pcv_conv2d_gated_fused": (_I, [_VP, ctypes.POINTER(ConvDesc), _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "pcv_fc_f32": (_I, [_VP, _VP, _VP, _VP, _VP, _I, _I, _I, _I, _VP]),
    "pcv_conv2d_maxpool_supported": (_I, [ctypes.POINTER(ConvDesc), _I, _I, _I, _I]),
    "pcv_conv2d_maxpool_fused": (_I, [_VP, ctypes.POINTER(ConvDesc), _VP, _VP, _VP, _VP, _VP, _I, _I, _I, _I, _VP]),
    "pcv_conv2d_nchw_stem_supported": (_I, [ctypes.POINTER(ConvDesc), _I]),
    "pcv_conv2d_nchw_stem_fused": (_I, [_VP, ctypes.POINTER(ConvDesc), _VP, _VP, _VP, _VP, _VP, _I, _VP]),
    "pcv_conv1x1_pair_supported": (_I, [ctypes.POINTER(ConvDesc), ctypes.POINTER(ConvDesc)]),
    "pcv_conv1x1_pair_fused": (_I, [_VP, ctypes.POINTER(ConvDesc), ctypes.POINTER(ConvDesc), _VP, _VP, _VP, _VP, _VP, _VP, _VP,
                                    _VP, _VP, _VP, _VP]),
    "pcv_conv1x1_pair_gated_supported": (_I, [ctypes.POINTER(ConvDesc), ctypes.POINTER(ConvDesc)]),
    "pcv_conv1x1_pair_gated_fused": (_I, [_VP, ctypes.POINTER(ConvDesc), ctypes.POINTER(ConvDesc), _VP, _VP, _VP, _VP, _VP, _VP, _VP,
                                          _VP, _VP, _VP, _VP, _VP]),
    "pcv_conv1x1_pair_idconv_supported": (_I, [ctypes.POINTER(ConvDesc), ctypes.POINTER(ConvDesc), ctypes.POINTER(ConvDesc)]),
    "pcv_conv1x1_pair_idconv_fused": (_I, [_VP, ctypes.POINTER(ConvDesc), ctypes.POINTER(ConvDesc), ctypes.POINTER(ConvDesc),
                                           _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "pcv_mbconv_supported": (_I, [ctypes.POINTER(ConvDesc), ctypes.POINTER(ConvDesc), ctypes.POINTER(ConvDesc)]),
    "pcv_mbconv_fused": (_I, [_VP, ctypes.POINTER(ConvDesc), ctypes.POINTER(ConvDesc), ctypes.POINTER(ConvDesc), _VP, _VP, _VP, _VP,
                              _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "pcv_bn_act": (_I, [_VP, _VP, _VP, _VP, _VP, ctypes.c_long, _I, _I, _I, _I, _VP]),
    "pcv_se_scale": (_I, [_VP, _VP, _VP, _VP, _VP, _I, _I, _I, _I, _I, _VP]),
}

_lib = None
_ctxs = {}
_lock = threading.RLock()


def lib():
    """The loaded shared library (raises if it has not been built)."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise RuntimeError(
                        "pytorchcv_amd: HIP extension {} is missing - build it with `python -c 'import __graft_entry__ as g; "
                        "g.build()'` or `make -C pytorchcv_amd/csrc`. There is no CPU fallback.".format(LIB_PATH))
                L = ctypes.CDLL(LIB_PATH)
                for name, (res, args) in _SIGS.items():
                    fn = getattr(L, name)          # AttributeError here = header/library mismatch
                    fn.restype = res
                    fn.argtypes = args
                if L.pcv_abi_version() != PCV_ABI_VERSION or L.pcv_conv_desc_size() != ctypes.sizeof(ConvDesc):
                    raise RuntimeError("pytorchcv_amd: {} has ABI version {} / a {}-byte pcv_conv_desc, this binding expects "
                                       "version {} / {} bytes - rebuild the library".format(
                                           LIB_PATH, L.pcv_abi_version(), L.pcv_conv_desc_size(), PCV_ABI_VERSION,
                                           ctypes.sizeof(ConvDesc)))
                _lib = L
    return _lib


def exported_symbols():
    return sorted(_SIGS.keys())


def ctx_for(device_index: int):
    """One library context per device, created on first use."""
    c = _ctxs.get(device_index)
    if c is None:
        with _lock:
            c = _ctxs.get(device_index)
            if c is None:
                L = lib()
                h = _VP()
                rc = L.pcv_create(ctypes.byref(h), int(device_index))
                if rc != 0:
                    raise PcvError(rc, (L.pcv_last_error(None) or b"").decode())
                c = h
                _ctxs[device_index] = c
    return c


def check(rc: int, ctx):
    if rc != 0:
        raise PcvError(rc, (lib().pcv_last_error(ctx) or b"").decode())
