"""CPU: AddressSanitizer + UndefinedBehaviorSanitizer builds of the two pieces of native code that run on the host
(sanitizers on the CPU build only - GPU ASan is not available on this pool):

  * oracle/cref.c (the plain-C restatement of the reference's ops, test infrastructure) - `make -C oracle libcref_asan.so`;
  * the planning half of the C ABI in pytorchcv_amd/csrc/pcv_api.hip compiled for the host alone
    (`make -C pytorchcv_amd/csrc libpcv_host_asan.so`): descriptor validation, packed-weight sizes, every `*_supported`
    predicate, context creation without a device. No kernel is launched (there is no GPU here).

Each runs in a child python with the sanitizer runtime preloaded; a report aborts the child (non-zero exit)."""

import os
import sys
import json
import subprocess
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pytorchcv_amd", "csrc")


def _child(preload, code, extra_env=None, timeout=300):
    env = dict(os.environ, LD_PRELOAD=preload, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", PYTHONPATH=ROOT)
    env.update(extra_env or {})
    return subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)


_CREF_CHILD = r"""
import sys, json, numpy as np
from oracle import cref
rng = np.random.default_rng(0)
out = {}
def rnd(*s): return rng.standard_normal(s).astype(np.float32)
# dense / strided / dilated / grouped / depthwise / asymmetric padding / 1x1 maps / kernel larger than the padded input edge cases
cases = [dict(x=(2, 8, 9, 7), w=(12, 8, 3, 3), stride=1, padding=1), dict(x=(1, 8, 9, 7), w=(8, 8, 3, 3), stride=2, padding=1, dilation=1),
         dict(x=(1, 4, 11, 11), w=(6, 4, 3, 3), stride=1, padding=2, dilation=2), dict(x=(2, 16, 6, 6), w=(16, 4, 3, 3), padding=1, groups=4),
         dict(x=(1, 12, 7, 5), w=(12, 1, 5, 5), padding=2, groups=12, stride=2), dict(x=(1, 3, 10, 10), w=(5, 3, 7, 7), stride=2, padding=(3, 2, 1, 0)),
         dict(x=(3, 5, 1, 1), w=(7, 5, 1, 1)), dict(x=(1, 2, 2, 2), w=(2, 2, 3, 3), padding=1)]
for i, c in enumerate(cases):
    x, w = rnd(*c["x"]), rnd(*c["w"])
    kw = {k: v for k, v in c.items() if k not in ("x", "w")}
    O = w.shape[0]
    bn = (rnd(O) + 2, rnd(O), rnd(O), np.abs(rnd(O)) + 0.5)
    for act in (None, "relu", "relu6", "sigmoid", "swish", "hsigmoid", "hswish"):
        y = cref.conv_block_c(x, w, bias=rnd(O), bn=bn, act=act, **kw)
        out["conv%d_%s" % (i, act)] = float(np.abs(y).sum())
y = cref.conv_block_c(rnd(1, 4, 5, 5), rnd(4, 4, 3, 3), padding=1, act="relu", residual=rnd(1, 4, 5, 5), post_act="relu")
out["res"] = float(np.abs(y).sum())
out["maxpool"] = float(cref.maxpool2d_c(rnd(2, 3, 9, 9), 3, 2, 1).sum())
out["avgpool"] = float(cref.avgpool2d_c(rnd(2, 3, 7, 7), 7, 1).sum())
out["linear"] = float(cref.linear_c(rnd(3, 10), rnd(4, 10)).sum())
out["se"] = float(cref.se_gate_c(rnd(2, 8, 4, 4), rnd(2, 8, 1, 1), rnd(2), rnd(8, 2, 1, 1), rnd(8)).sum())
print(json.dumps(out))
"""


def test_c_oracle_under_asan_ubsan():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "libcref.so", "libcref_asan.so"])
    rt = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    if not os.path.isabs(rt) or not os.path.exists(rt):
        pytest.skip("gcc's libasan.so not found")
    san = _child(rt, _CREF_CHILD, {"ORACLE_CREF_LIB": os.path.join(ROOT, "oracle", "libcref_asan.so")})
    assert san.returncode == 0, "sanitizer report:\n" + san.stderr[-4000:]
    plain = subprocess.run([sys.executable, "-c", _CREF_CHILD], env=dict(os.environ, PYTHONPATH=ROOT), stdout=subprocess.PIPE, text=True)
    a, b = json.loads(san.stdout.strip().splitlines()[-1]), json.loads(plain.stdout.strip().splitlines()[-1])
    assert a.keys() == b.keys() and len(a) >= 60
    for k in a:                                           # -O1 vs -O2: the same arithmetic up to contraction
        assert abs(a[k] - b[k]) <= 1e-4 * max(1.0, abs(b[k])), k


_PLAN_CHILD = r"""
import ctypes, importlib.util, json, os, sys
spec = importlib.util.spec_from_file_location("pcv_lib_binding", os.path.join(sys.argv[1], "pytorchcv_amd", "_lib.py"))
B = importlib.util.module_from_spec(spec); spec.loader.exec_module(B)          # the ctypes binding alone: no torch in this process
L = ctypes.CDLL(sys.argv[2])
for name, (res, args) in B._SIGS.items():
    fn = getattr(L, name); fn.restype = res; fn.argtypes = args
D = B.ConvDesc
assert L.pcv_abi_version() == B.PCV_ABI_VERSION and L.pcv_conv_desc_size() == ctypes.sizeof(D)
def desc(**kw):
    base = dict(N=2, H=14, W=14, Cin=64, Cout=64, kh=3, kw=3, stride_h=1, stride_w=1, pad_t=1, pad_l=1, pad_b=1, pad_r=1, dil_h=1, dil_w=1,
                groups=1, act=1, post_act=0, has_residual=0, dtype=1, out_dtype=1, x_cpitch=0, x_wpitch=0, y_cpitch=0)
    base.update(kw)
    if base["x_cpitch"] == 0: base["x_cpitch"] = base["Cin"]
    if base["x_wpitch"] == 0: base["x_wpitch"] = base["W"]
    return D(**base)
n = ctypes.c_size_t(0)
seen = {"ok": 0, "refused": 0}
def size(d, dw=False):
    rc = (L.pcv_dwconv_packed_bytes if dw else L.pcv_conv_packed_bytes)(ctypes.byref(d), ctypes.byref(n))
    seen["ok" if rc == 0 else "refused"] += 1
    return rc
BIG = 2 ** 31 - 1
# every convolution family of the path, in all three storage types
for dt in (0, 1, 2):
    for kw in (dict(), dict(kh=1, kw=1, pad_t=0, pad_l=0, pad_b=0, pad_r=0, Cin=256, Cout=1024), dict(stride_h=2, stride_w=2),
               dict(dil_h=2, dil_w=2, pad_t=2, pad_l=2, pad_b=2, pad_r=2), dict(groups=32, Cin=128, Cout=128), dict(groups=32, Cin=1024, Cout=1024, stride_h=2, stride_w=2),
               dict(Cin=3, Cout=64, kh=7, kw=7, stride_h=2, stride_w=2, pad_t=3, pad_l=3, pad_b=3, pad_r=3, H=224, W=224, x_cpitch=4, x_wpitch=224),
               dict(Cin=2048, Cout=1000, kh=1, kw=1, pad_t=0, pad_l=0, pad_b=0, pad_r=0, H=1, W=1, out_dtype=0),
               dict(Cin=24, Cout=40, kh=5, kw=5, pad_t=2, pad_l=2, pad_b=2, pad_r=2), dict(kh=1, kw=7, pad_t=0, pad_b=0, pad_l=3, pad_r=3)):
        d = desc(dtype=dt, out_dtype=kw.pop("out_dtype", dt), **kw)
        assert size(d) == 0 and n.value > 0, kw
    for k in (3, 5):
        for s in (1, 2):
            d = desc(dtype=dt, out_dtype=dt, Cin=96, Cout=96, groups=96, kh=k, kw=k, stride_h=s, stride_w=s, pad_t=k // 2, pad_l=k // 2, pad_b=k // 2, pad_r=k // 2)
            taps = (k * k * 96 * (4 if dt == 0 else 2) + 15) // 16 * 16
            assert size(d, dw=True) == 0 and n.value == taps + (3 * 6 * 1024 if (k == 3 and dt != 0) else 0)      # 16-bit 3x3: + the sparse-MFMA fragments
# refused, never read out of bounds or overflowed: bad counts, sizes at the edge of int32, a stale struct_size, unknown dtypes
bad = [dict(Cin=0), dict(Cout=-8), dict(kh=0), dict(kh=16, kw=16), dict(stride_h=0), dict(dil_w=-1), dict(groups=0), dict(groups=3), dict(dtype=7), dict(out_dtype=2, dtype=1),
       dict(x_cpitch=32), dict(Cin=BIG, Cout=BIG), dict(Cin=BIG - 6, Cout=8), dict(Cin=8, Cout=BIG - 6), dict(N=BIG, H=BIG, W=BIG), dict(kh=15, kw=15, Cin=1 << 24, Cout=1 << 20),
       dict(pad_t=BIG, pad_b=BIG), dict(dil_h=BIG), dict(groups=BIG, Cin=BIG, Cout=BIG)]
for kw in bad:
    d = desc(**kw)
    rc = size(d)
    assert rc in (0, -1), (kw, rc)
d = desc(); d.struct_size = 96
assert size(d) == -1
assert L.pcv_conv_packed_bytes(None, ctypes.byref(n)) == -1 and L.pcv_conv_packed_bytes(ctypes.byref(desc()), None) == -1
assert L.pcv_dwconv_packed_bytes(ctypes.byref(desc()), ctypes.byref(n)) == -1          # not depthwise
# the *_supported predicates on covered, uncovered and nonsensical descriptors (NULL included)
stem = desc(Cin=3, Cout=64, kh=7, kw=7, stride_h=2, stride_w=2, pad_t=3, pad_l=3, pad_b=3, pad_r=3, H=224, W=224, x_cpitch=4, x_wpitch=224)
assert L.pcv_conv2d_maxpool_supported(ctypes.byref(stem), 3, 2, 1, 0) == 1 and L.pcv_conv2d_maxpool_supported(ctypes.byref(stem), 2, 2, 0, 0) == 0
assert L.pcv_conv2d_maxpool_supported(None, 3, 2, 1, 0) == 0 and L.pcv_conv2d_maxpool_supported(ctypes.byref(desc()), 3, 2, 1, 0) == 0
assert L.pcv_conv2d_nchw_stem_supported(ctypes.byref(stem), 1) == 1 and L.pcv_conv2d_nchw_stem_supported(ctypes.byref(desc()), 0) == 0
def c1(cin, cout, res=0, act=1, **kw):
    kw.setdefault("H", 28); kw.setdefault("W", 28)
    return desc(Cin=cin, Cout=cout, kh=1, kw=1, pad_t=0, pad_l=0, pad_b=0, pad_r=0, has_residual=res, act=act, **kw)
pairs = [(c1(64, 256, 1, 0, post_act=1), c1(256, 64), 1), (c1(128, 512, 1, 0, post_act=1), c1(512, 128), 1), (c1(256, 1024, 1, 0, post_act=1), c1(1024, 256), 1),
         (c1(512, 2048, 1, 0, post_act=1), c1(2048, 512), 0), (c1(64, 256, 0, 0), c1(256, 64), 0), (c1(64, 256, 1, 0, dtype=0, out_dtype=0), c1(256, 64, dtype=0, out_dtype=0), 0),
         (c1(64, 256, 1, 0, N=BIG), c1(256, 64, N=BIG), 0)]
for a, b, want in pairs:
    assert L.pcv_conv1x1_pair_supported(ctypes.byref(a), ctypes.byref(b)) == want
    L.pcv_conv1x1_pair_gated_supported(ctypes.byref(a), ctypes.byref(b))
    L.pcv_conv1x1_pair_idconv_supported(ctypes.byref(c1(64, 256, 0, 0)), ctypes.byref(a), ctypes.byref(b))
assert L.pcv_conv1x1_pair_supported(None, None) == 0 and L.pcv_conv1x1_pair_idconv_supported(None, None, None) == 0
def unit(cin, cmid, cout, s, H=56, expand=True):
    e = c1(cin, cmid, act=2, H=H, W=H) if expand else None
    dd = desc(Cin=cmid, Cout=cmid, groups=cmid, stride_h=s, stride_w=s, act=2, H=H, W=H)
    Ho = (H - 1) // s + 1
    p = c1(cmid, cout, act=0, H=Ho, W=Ho)
    return e, dd, p
for args, want in (((24, 144, 24, 1), 1), ((16, 96, 24, 2, 112), 1), ((96, 576, 96, 1, 14), 1), ((160, 960, 160, 1, 7), 0), ((24, 144, 24, 1, 6), 0),
                   ((32, 32, 16, 1, 112, False), 1), ((BIG, BIG, 8, 1), 0)):
    e, dd, p = unit(*args)
    assert L.pcv_mbconv_supported(ctypes.byref(e) if e is not None else None, ctypes.byref(dd), ctypes.byref(p)) == want, args
# no device here: creation fails cleanly with a message, every entry point refuses a NULL context
h = ctypes.c_void_p()
rc = L.pcv_create(ctypes.byref(h), 0)
assert rc in (-3, -2) and h.value is None and len(L.pcv_last_error(None)) > 0, rc
assert L.pcv_create(None, 0) == -1 and L.pcv_destroy(None) == 0
assert L.pcv_conv2d_fused(None, ctypes.byref(desc()), None, None, None, None, None, None, None) == -1
assert L.pcv_fp16_guard_begin(None, None, None) == -1 and L.pcv_set_tuning(None, b"d3x3", 0) == -1
print(json.dumps(seen))
"""


def test_abi_planner_under_asan_ubsan():
    subprocess.check_call(["make", "-s", "-j8", "-C", CSRC, "libpcv_host_asan.so"])   # (17 translation units at -O1 -g: ~10 min serially)
    clang = "/opt/rocm/lib/llvm/bin/clang"
    rt = subprocess.check_output([clang, "-print-file-name=libclang_rt.asan-x86_64.so"], text=True).strip()
    if not os.path.exists(rt):
        pytest.skip("clang's shared ASan runtime not found")
    p = subprocess.run([sys.executable, "-c", _PLAN_CHILD, ROOT, os.path.join(CSRC, "libpcv_host_asan.so")],
                       env=dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1",
                                UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1"),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode == 0, "sanitizer report / failed check:\n" + p.stderr[-4000:]
    seen = json.loads(p.stdout.strip().splitlines()[-1])
    assert seen["ok"] >= 40 and seen["refused"] >= 10
