"""CPU: the drop-in boundary. The C-ABI library loads and exports every symbol include/pcv_amd.h declares (no compute
calls without a GPU), the pure-host planning entry points behave, and the Python mirror of the reference interface
(get_model, state_dict layout, parameter counts, weight store) matches the reference's known answers."""

import os
import re
import ctypes
import hashlib
import pytest
import torch
import util
from cases import PARAM_COUNTS, KEY_COUNTS

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    text = open(os.path.join(ROOT, "include", "pcv_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pcv_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from pytorchcv_amd import _lib
    L = _lib.lib()
    declared = _header_functions()
    assert len(declared) >= 19
    for name in declared:
        assert hasattr(L, name), "libpcv_amd.so does not export {}".format(name)
    assert sorted(_lib.exported_symbols()) == declared, "ctypes signature table and header disagree"
    assert L.pcv_abi_version() == _lib.PCV_ABI_VERSION == _header_abi_version()


def _header_abi_version():
    text = open(os.path.join(ROOT, "include", "pcv_amd.h")).read()
    return int(re.search(r"#define\s+PCV_ABI_VERSION\s+(\d+)", text).group(1))


def _header_desc_fields():
    text = open(os.path.join(ROOT, "include", "pcv_amd.h")).read()
    body = text[text.index("typedef struct pcv_conv_desc {"):text.index("} pcv_conv_desc;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in re.findall(r"int32_t\s+([^;]+);", body):
        fields += [f.strip() for f in decl.split(",")]
    return fields


def test_conv_desc_matches_header_layout():
    from pytorchcv_amd import _lib
    from pytorchcv_amd._lib import ConvDesc
    fields = _header_desc_fields()
    assert fields[0] == "struct_size" and fields[-1] == "y_cpitch"
    assert fields == [f[0] for f in ConvDesc._fields_]
    assert ctypes.sizeof(ConvDesc) == 4 * len(fields) == _lib.lib().pcv_conv_desc_size()
    assert ConvDesc().struct_size == ctypes.sizeof(ConvDesc)            # filled in by the constructor


def test_integration_stub_matches_header():
    """INTEGRATION.md's reference-side ctypes stub declares exactly the header's descriptor (a stub one field short made the
    library read the output pitch past the caller's struct in ABI version 1) and asserts the ABI version it was written for."""
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    stub = md[md.index("class ConvDesc(ctypes.Structure):"):]
    stub = stub[:stub.index(")]") + 2]
    names = re.findall(r'"([A-Za-z_0-9]+)"', stub)
    assert names == _header_desc_fields()
    m = re.search(r"pcv_abi_version\(\) == (\d+) and L.pcv_conv_desc_size\(\) == ctypes.sizeof\(ConvDesc\)", md)
    assert m and int(m.group(1)) == _header_abi_version()
    call = md[md.index("d = ConvDesc("):]
    call = call[:call.index("n = ctypes.c_size_t()")]
    used = re.findall(r"\b([A-Za-z_0-9]+)=", call)
    assert sorted(used) == sorted(_header_desc_fields()), "the stub's example call must set every field"


def test_stale_descriptor_is_refused():
    """A descriptor whose struct_size is not this library's sizeof(pcv_conv_desc) (a binding built against another layout)
    is rejected by the host-side planning entry points instead of being read."""
    from pytorchcv_amd import _lib
    L = _lib.lib()
    n = ctypes.c_size_t(0)
    d = _desc()
    assert L.pcv_conv_packed_bytes(ctypes.byref(d), ctypes.byref(n)) == 0
    for bad in (0, ctypes.sizeof(_lib.ConvDesc) - 4, ctypes.sizeof(_lib.ConvDesc) + 4):
        d.struct_size = bad
        assert L.pcv_conv_packed_bytes(ctypes.byref(d), ctypes.byref(n)) != 0
        assert L.pcv_dwconv_packed_bytes(ctypes.byref(d), ctypes.byref(n)) != 0
        assert L.pcv_conv1x1_pair_supported(ctypes.byref(d), ctypes.byref(d)) == 0
        assert L.pcv_conv2d_maxpool_supported(ctypes.byref(d), 3, 2, 1, 0) == 0
        assert L.pcv_mbconv_supported(None, ctypes.byref(d), ctypes.byref(d)) == 0


def test_pytorchcv_import_path_is_the_drop_in():
    """`from pytorchcv.model_provider import get_model` (reference model_provider.py:1364-1382) works unchanged and resolves
    to this package: same registry, same classes, same weight store."""
    import pytorchcv
    from pytorchcv.model_provider import get_model
    import pytorchcv_amd.model_provider as amd_mp
    assert get_model is amd_mp.get_model
    import pytorchcv.models.resnet as r
    import pytorchcv_amd.models.resnet as ar
    assert r is ar and r.resnet50 is ar.resnet50
    from pytorchcv.models.common.model_store import load_model, calc_net_weight_count, get_model_file   # noqa: F401
    import pytorchcv_amd.models.common.model_store as ams
    assert load_model is ams.load_model
    from pytorchcv.models.common.conv import ConvBlock, conv3x3_block                      # noqa: F401
    with pytest.raises(ValueError, match="Unsupported model: nosuchnet"):
        get_model("NoSuchNet")
    net = get_model("resnet18")
    assert calc_net_weight_count(net) == PARAM_COUNTS["resnet18"]
    with pytest.raises(ImportError):
        import pytorchcv.models.no_such_family                                            # noqa: F401
    assert pytorchcv.set_compute_dtype is not None


def _desc(**kw):
    from pytorchcv_amd._lib import ConvDesc
    base = dict(N=1, H=8, W=8, Cin=64, Cout=64, kh=3, kw=3, stride_h=1, stride_w=1, pad_t=1, pad_l=1, pad_b=1, pad_r=1,
                dil_h=1, dil_w=1, groups=1, act=1, post_act=0, has_residual=0, dtype=1, out_dtype=1, x_cpitch=64, x_wpitch=8)
    base.update(kw)
    return ConvDesc(**base)


def test_packed_size_planning_is_pure_host_logic():
    from pytorchcv_amd import _lib
    L = _lib.lib()
    n = ctypes.c_size_t(0)
    # 3x3, 64->64, bf16: K = 9*64 = 576 -> 9 K-steps of 64; rows padded to 64; table 9*8 chunks * 8 B -> 768 B (256-aligned)
    assert L.pcv_conv_packed_bytes(ctypes.byref(_desc()), ctypes.byref(n)) == 0
    assert n.value == 768 + 64 * 576 * 2
    # fp32: chunk = 4 elements, K-step = 32 -> 18 K-steps
    assert L.pcv_conv_packed_bytes(ctypes.byref(_desc(dtype=0, out_dtype=0)), ctypes.byref(n)) == 0
    assert n.value == 1280 + 64 * 576 * 4          # 18 K-steps x 64 B of table, rounded up to 256
    # padded 4-channel stem, 7x7/2 pad 3: 7 rows x 4 pixel pairs = 28 chunks -> 4 K-steps, K = 256
    d = _desc(Cin=3, Cout=64, kh=7, kw=7, stride_h=2, stride_w=2, pad_t=3, pad_l=3, pad_b=3, pad_r=3, x_cpitch=4, H=32, W=32,
              x_wpitch=32)
    assert L.pcv_conv_packed_bytes(ctypes.byref(d), ctypes.byref(n)) == 0
    assert n.value == 7 * 64 * 32 * 2            # Cout <= 64: dedicated stem kernel, blob = [kh][64][8 pixels x 4 channels]
    d.Cout = 128                                 # wider stems stay on the implicit GEMM with the pixel-pair K-chunk table
    assert L.pcv_conv_packed_bytes(ctypes.byref(d), ctypes.byref(n)) == 0
    assert n.value == 256 + 128 * 256 * 2
    # grouped g=32, 4 channels per group -> 4 group blocks of 32 channels, K = 9*32 = 288 -> 5 K-steps (320)
    d = _desc(Cin=128, Cout=128, groups=32, x_cpitch=128)
    assert L.pcv_conv_packed_bytes(ctypes.byref(d), ctypes.byref(n)) == 0
    # ... followed by the second blob for gconv3x3_kernel: [128 / 16 slabs][5 K-steps][16 rows][32 K] bf16
    assert n.value == 512 + 4 * 32 * 320 * 2 + 8 * 5 * 16 * 32 * 2
    d = _desc(Cin=128, Cout=128, groups=32, x_cpitch=128, stride_h=2, stride_w=2)      # stride 2 (gconv3x3r_kernel): the same blob
    assert L.pcv_conv_packed_bytes(ctypes.byref(d), ctypes.byref(n)) == 0
    assert n.value == 512 + 4 * 32 * 320 * 2 + 8 * 5 * 16 * 32 * 2
    for stride in (1, 2):                        # 32 channels per group: [128 / 16 slabs][9 taps][16 rows][32 input channels]
        d = _desc(Cin=128, Cout=128, groups=4, x_cpitch=128, stride_h=stride, stride_w=stride)
        assert L.pcv_conv_packed_bytes(ctypes.byref(d), ctypes.byref(n)) == 0
        assert n.value == 512 + 4 * 32 * 320 * 2 + 8 * 9 * 16 * 32 * 2
    d = _desc(Cin=128, Cout=128, groups=32, x_cpitch=128, stride_h=2, stride_w=1)      # anisotropic stride: generic layout only
    assert L.pcv_conv_packed_bytes(ctypes.byref(d), ctypes.byref(n)) == 0
    assert n.value == 512 + 4 * 32 * 320 * 2
    # depthwise goes through its own entry point
    d = _desc(Cin=144, Cout=144, groups=144, x_cpitch=144)
    assert L.pcv_dwconv_packed_bytes(ctypes.byref(d), ctypes.byref(n)) == 0
    # the taps [9][144] (rounded up to 16 bytes), then - 16-bit 3x3 - the compressed diagonal fragments of the sparse matrix instruction for
    # the fused inverted-residual kernel: 5 chunks of 32 channels x 6 fragments x 1 KB (csrc/aux_kernels.hpp pack_dw_sparse_kernel)
    assert n.value == 9 * 144 * 2 + 5 * 6 * 1024
    d = _desc(Cin=144, Cout=144, groups=144, x_cpitch=144, dtype=0, out_dtype=0)       # fp32: the taps only
    assert L.pcv_dwconv_packed_bytes(ctypes.byref(d), ctypes.byref(n)) == 0
    assert n.value == 9 * 144 * 4
    d = _desc(Cin=40, Cout=40, groups=40, x_cpitch=40, kh=5, kw=5, pad_t=2, pad_b=2, pad_l=2, pad_r=2)       # 5x5: the taps only (25 x 40 x 2 = 2000 bytes)
    assert L.pcv_dwconv_packed_bytes(ctypes.byref(d), ctypes.byref(n)) == 0
    assert n.value == 2000


@pytest.mark.parametrize("bad", [dict(groups=3), dict(kh=16), dict(dtype=7), dict(out_dtype=2), dict(x_cpitch=60),
                                 dict(stride_h=0), dict(Cin=3, x_cpitch=4, stride_w=1)])
def test_unsupported_configurations_are_rejected_not_guessed(bad):
    from pytorchcv_amd import _lib
    n = ctypes.c_size_t(0)
    assert _lib.lib().pcv_conv_packed_bytes(ctypes.byref(_desc(**bad)), ctypes.byref(n)) != 0


def test_no_gpu_means_loud_failure_not_fallback():
    from pytorchcv_amd import _lib
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.PcvError, match="no HIP device"):
        _lib.ctx_for(0)
    from pytorchcv_amd.model_provider import get_model
    net = get_model("resnet18").eval()
    with pytest.raises(RuntimeError, match="MI355X"):
        net(torch.zeros(1, 3, 224, 224))


def test_product_never_imports_the_oracle():
    for base, _, files in os.walk(os.path.join(ROOT, "pytorchcv_amd")):
        for f in files:
            if f.endswith(".py"):
                text = open(os.path.join(base, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), "{} imports oracle".format(f)


# ---- reference interface mirror -------------------------------------------------------------------------------
def test_get_model_contract():
    from pytorchcv_amd.model_provider import get_model
    with pytest.raises(ValueError, match="Unsupported model: nosuchnet"):
        get_model("NoSuchNet")
    net = get_model("ResNet18")                      # case-insensitive (model_provider.py:1378)
    assert net.in_size == (224, 224) and net.num_classes == 1000
    assert hasattr(net, "features") and hasattr(net, "output")
    with pytest.raises(ValueError, match="Pretrained model for"):
        from pytorchcv_amd.models.resnet import get_resnet
        get_resnet(blocks=18, model_name="not_in_index", pretrained=True)
    with pytest.raises(ValueError, match="model_name"):
        get_resnet(blocks=18, pretrained=True)
    with pytest.raises(ValueError, match="Unsupported ResNet with number of blocks: 19"):
        get_resnet(blocks=19)


def test_auto_dtype_is_resolved_once_per_family():
    """ "auto" resolves to the family's 16-bit mode on EVERY sub-module (`net.features(x)` or a unit called with an NCHW
    tensor ran an fp16-family backbone in bf16): every module of a net reports the type the net itself reports."""
    from pytorchcv_amd import engine
    from pytorchcv_amd.model_provider import get_model
    for name, want in (("mobilenetv2_w1", "fp16"), ("mobilenetv3_small_w1", "fp16"), ("efficientnet_b0", "fp16"), ("resnet18", "bf16"),
                       ("preresnet18", "bf16")):
        net = get_model(name).eval()
        if os.environ.get("PCV_AMD_DTYPE", "auto") == "auto":
            assert engine.compute_dtype_of(net) == want
        kinds = {engine.compute_dtype_of(m) for m in net.modules()}
        assert kinds == {engine.compute_dtype_of(net)}, (name, kinds)
        engine.set_compute_dtype(net, "bf16")
        assert {engine.compute_dtype_of(m) for m in net.modules()} == {"bf16"}


@pytest.mark.parametrize("name", util.MODELS)
def test_param_counts_and_state_dict_manifest(name):
    """Known answers of the reference's own asserts + the key/shape/dtype manifest captured from the reference."""
    from pytorchcv_amd.model_provider import get_model
    from pytorchcv_amd.models.common.model_store import calc_net_weight_count, get_model_weight_count
    net = get_model(name).eval()
    man = util.model_manifest(name)
    assert calc_net_weight_count(net) == PARAM_COUNTS[name] == man["param_count"] == get_model_weight_count(name)
    sd = net.state_dict()
    assert list(sd.keys()) == list(man["keys"].keys())
    if name in KEY_COUNTS:
        assert len(sd) == KEY_COUNTS[name]
    for k, v in sd.items():
        assert list(v.shape) == man["keys"][k][0] and str(v.dtype) == "torch." + man["keys"][k][1], k
    # the fixture state dict (generated against the reference's manifest) loads strictly
    net.load_state_dict(util.model_state(name), strict=True)


def test_more_variants_match_csv_param_counts():
    from pytorchcv_amd.model_provider import get_model
    from pytorchcv_amd.models.common.model_store import calc_net_weight_count, get_model_metainfo_dict
    table = get_model_metainfo_dict()
    for name in ["resnet10", "resnet34", "resnet50b", "resnetbc26b", "mobilenetv2_wd2", "mobilenetv2_w3d4", "resnext26_32x4d",
                 "resnext14_16x4d", "seresnet18", "seresnetbc26b"]:
        assert calc_net_weight_count(get_model(name)) == table[name][0], name


def test_weight_store_local_file_roundtrip(tmp_path, monkeypatch):
    """`pretrained=True` semantics without network: a correctly named, SHA-1-matching local file is loaded; extra keys are
    ignored; a corrupt file triggers the download branch (which must fail here, not load garbage)."""
    from pytorchcv_amd.models.common import model_store as ms
    from pytorchcv_amd.model_provider import get_model
    net = get_model("resnet10").eval()
    sd = util.synth_state_dict(net.state_dict(), seed=7)
    sd_extra = dict(sd)
    sd_extra["not.a.key"] = torch.zeros(1)
    path = tmp_path / "w.pth"
    torch.save(sd_extra, str(path))
    sha1 = hashlib.sha1(open(path, "rb").read()).hexdigest()
    monkeypatch.setattr(ms, "get_model_metainfo_dict", lambda: {"resnet10": (5418792, "1253", sha1, "v0.0.0")})
    final = tmp_path / "resnet10-1253-{}.pth".format(sha1[:8])
    os.rename(path, final)
    assert ms.get_model_file("resnet10", str(tmp_path)) == str(final)
    net2 = get_model("resnet10", pretrained=True, root=str(tmp_path)).eval()
    for k, v in net2.state_dict().items():
        assert torch.equal(v, sd[k]), k
    with open(final, "ab") as f:
        f.write(b"x")
    monkeypatch.setattr(ms, "_fetch", lambda url, p, retries=5: (_ for _ in ()).throw(RuntimeError("no network")))
    with pytest.raises(RuntimeError, match="no network"):
        ms.get_model_file("resnet10", str(tmp_path))


def test_synth_is_bit_stable():
    from pytorchcv_amd.synth import hash_uniform, hash_normal, synth_input
    u = hash_uniform(1234, 42, 4)
    n = hash_normal(0, 0x1A9E0000, 3)
    assert [float(v) for v in u] == [float(v) for v in hash_uniform(1234, 42, 4)]
    assert hashlib.sha1(synth_input(1, 3, 8, 8, seed=0).numpy().tobytes()).hexdigest() == \
        hashlib.sha1(synth_input(1, 3, 8, 8, seed=0).numpy().tobytes()).hexdigest()
    # frozen known answers: a change here silently invalidates every golden fixture
    assert [float(v).hex() for v in u] == ['0x1.17f67e0000000p-1', '0x1.ff3ff40000000p-2', '0x1.d3dae80000000p-1', '0x1.6febca0000000p-1']
    assert [float(v).hex() for v in n] == ['-0x1.a49a395560262p-1', '-0x1.09dee97fa3aa4p-2', '-0x1.e31e97057927cp-1']
    g, ids = util.model_golden("resnet18")
    assert ids == [16, 24, 31, 44] and g.shape == (4, 1000)
    assert len(n) == 3


def test_weight_store_download_extracts_only_the_expected_member(tmp_path, monkeypatch):
    """The download branch with a stand-in for the HTTP fetch: only the expected `.pth` leaves the archive (a path-traversal
    member and a stray file stay inside it), it becomes the model file only after its SHA-1 matched, and a wrong hash leaves
    nothing behind."""
    import zipfile
    from pytorchcv_amd.models.common import model_store as ms
    from pytorchcv_amd.model_provider import get_model
    net = get_model("resnet10").eval()
    sd = util.synth_state_dict(net.state_dict(), seed=8)
    src = tmp_path / "src.pth"
    torch.save(sd, str(src))
    blob = open(src, "rb").read()
    sha1 = hashlib.sha1(blob).hexdigest()
    name = "resnet10-1253-{}.pth".format(sha1[:8])
    store = tmp_path / "store"
    urls = []

    def fake_fetch(url, path, retries=5):
        urls.append(url)
        with zipfile.ZipFile(path, "w") as zf:
            zf.writestr("../evil.txt", b"escaped")
            zf.writestr("stray.bin", b"stray")
            zf.writestr(name, blob)
        return path
    monkeypatch.setattr(ms, "_fetch", fake_fetch)
    monkeypatch.setattr(ms, "get_model_metainfo_dict", lambda: {"resnet10": (5418792, "1253", sha1, "v0.0.9")})
    got = ms.get_model_file("resnet10", str(store))
    assert got == str(store / name) and urls and urls[0].endswith("/releases/download/v0.0.9/{}.zip".format(name))
    assert sorted(os.listdir(store)) == [name]                    # nothing else was unpacked, the zip is gone
    assert not (tmp_path / "evil.txt").exists()
    ms.load_model(net, got)
    for k, v in net.state_dict().items():
        assert torch.equal(v, sd[k]), k
    # wrong hash in the table: refused, no partial file kept
    os.remove(got)
    monkeypatch.setattr(ms, "get_model_metainfo_dict", lambda: {"resnet10": (5418792, "1253", sha1[:8] + "0" * 32, "v0.0.9")})
    with pytest.raises(ValueError, match="different hash"):
        ms.get_model_file("resnet10", str(store))
    assert os.listdir(store) == []
