"""Dev tool (diagnostic build: MBR_EXTRA=-DMBR_CYCLES tests/tools/sh/mbr_variants.sh 0, run through ab_lib.py): shader cycles of
mbr_kernel per wave - block prologue (weights -> LDS), chunk loop and epilogue of the wave's second tile.
Usage: python tests/tools/ab_lib.py pytorchcv_amd/csrc/ab/libpcv_amd_mbr0.so tests/tools/mbr_cycles.py [Cin:Cout:H:expand[:stride] ...]"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, pytorchcv_amd
from pytorchcv_amd import engine, _lib
from pytorchcv_amd.models.mobilenetv2 import LinearBottleneck
from pytorchcv_amd.models.common.conv import mbconv_chain
from pytorchcv_amd.models.common.activ import create_activation_layer
from pytorchcv_amd.synth import synth_state_dict
dev = torch.device("cuda", 0); ctx = _lib.ctx_for(0)
def tune(k, v): _lib.check(_lib.lib().pcv_set_tuning(ctx, k.encode(), ctypes.c_int(v).value), ctx)
for spec in (sys.argv[1:] or ["32:16:112:0", "24:24:56:1", "32:32:28:1"]):
    Cin, Cout, H, expand, stride = (list(int(v) for v in spec.split(":")) + [1])[:5]
    dbg = torch.zeros(256 * 8 * 4, dtype=torch.int32, device=dev)
    ptr = dbg.data_ptr(); tune("dbg_lo", ctypes.c_int32(ptr & 0xFFFFFFFF).value); tune("dbg_hi", ctypes.c_int32(ptr >> 32).value)
    unit = LinearBottleneck(in_channels=Cin, out_channels=Cout, stride=stride, expansion=bool(expand), remove_exp_conv=False,
                            activation=(lambda: create_activation_layer("relu6"))).eval()
    unit.load_state_dict(synth_state_dict(unit.state_dict(), seed=3))
    unit = pytorchcv_amd.set_compute_dtype(unit.to(dev), "fp16")
    x = engine.NHWC(torch.randn(512, H, H, Cin, device=dev).to(torch.float16), 512, H, H, Cin)
    with torch.no_grad():
        for _ in range(10): mbconv_chain(unit.conv1, unit.conv2, unit.conv3, x, residual=x if unit.residual else None)
    torch.cuda.synchronize()
    d = dbg.cpu().view(-1, 4).to(torch.int64) & 0xFFFFFFFF
    d = d[d[:, 3] > 0].float()
    print(spec, "chunks", unit.conv2.conv.weight.shape[0] // 32 + (unit.conv2.conv.weight.shape[0] % 32 > 0), "records", len(d))
    for i, name in enumerate(("prologue", "chunk loop", "epilogue")):
        if len(d): print("  %-12s median %7.0f cycles   min %7.0f   max %7.0f" % (name, float(d[:, i].median()), float(d[:, i].min()), float(d[:, i].max())))
    tune("dbg_lo", 0); tune("dbg_hi", 0)
