"""Dev tool (diagnostic build: tests/tools/sh/kernel_variants.sh d1i_16bit cyc -DD1I_CYCLES, run through tests/tools/ab_lib.py): shader cycles of
d1i_kernel per wave - prologue (first slices -> LDS), K loop, epilogue + store drain."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, pytorchcv_amd
from pytorchcv_amd import engine, _lib
from pytorchcv_amd.models.common.conv import conv1x1_block
from pytorchcv_amd.synth import synth_state_dict
dev = torch.device("cuda", 0); ctx = _lib.ctx_for(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
C = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
Co = int(sys.argv[3]) if len(sys.argv) > 3 else 512
HW = int(sys.argv[4]) if len(sys.argv) > 4 else 14
dbg = torch.zeros(4096 * 4 * 8, dtype=torch.int32, device=dev)
def tune(k, v): _lib.check(_lib.lib().pcv_set_tuning(ctx, k.encode(), ctypes.c_int(v).value), ctx)
ptr = dbg.data_ptr(); tune("dbg_lo", ctypes.c_int32(ptr & 0xFFFFFFFF).value); tune("dbg_hi", ctypes.c_int32(ptr >> 32).value)
blk = conv1x1_block(in_channels=C, out_channels=Co).eval()
blk.load_state_dict(synth_state_dict(blk.state_dict(), seed=1))
blk = pytorchcv_amd.set_compute_dtype(blk.to(dev), "bf16")
x = engine.NHWC(torch.randn(N, HW, HW, C, device=dev).to(torch.bfloat16), N, HW, HW, C)
tune("d1i", 1)
with torch.no_grad():
    for _ in range(10): blk(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): blk(x)
    e1.record(); torch.cuda.synchronize()
print("launch %.1f us" % (e0.elapsed_time(e1) * 100))
d = dbg.cpu().view(-1, 8).to(torch.int64) & 0xFFFFFFFF
d = d[d[:, 7] > 0].float()
for i, name in enumerate(("prologue", "K loop", "epilogue + drain")):
    print("%-16s median %7.0f cycles   min %7.0f   max %7.0f" % (name, float(d[:, i].median()), float(d[:, i].min()), float(d[:, i].max())))
tot, rt = d[:, :3].sum(1), d[:, 5]
print("block total median %.0f cycles (%d wave records) in %.2f us of real time: in-kernel clock %.2f GHz; first to last block start %.2f us" % (
    float(tot.median()), len(d), float(rt.median()) / 100, float((tot / rt).median()) / 10, float(d[:, 6].max() - d[:, 6].min()) / 100))
tune("dbg_lo", 0); tune("dbg_hi", 0); tune("d1i", -1)
