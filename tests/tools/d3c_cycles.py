"""Dev tool (diagnostic build: make -C pytorchcv_amd/csrc EXTRA=-DD3C_CYCLES): shader cycles of one tile of d3c_kernel per wave -
K loop (252 MFMAs = 4 032 matrix-pipe cycles), epilogue, wait for the next patch + barrier."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, pytorchcv_amd
from pytorchcv_amd import engine, _lib
from pytorchcv_amd.models.common.conv import conv3x3_block
from pytorchcv_amd.synth import synth_state_dict
dev = torch.device("cuda", 0); ctx = _lib.ctx_for(0)
dbg = torch.zeros(256 * 4 * 4, dtype=torch.int32, device=dev)
def tune(k, v): _lib.check(_lib.lib().pcv_set_tuning(ctx, k.encode(), ctypes.c_int(v).value), ctx)
ptr = dbg.data_ptr(); tune("dbg_lo", ctypes.c_int32(ptr & 0xFFFFFFFF).value); tune("dbg_hi", ctypes.c_int32(ptr >> 32).value)
blk = conv3x3_block(in_channels=64, out_channels=64).eval()
blk.load_state_dict(synth_state_dict(blk.state_dict(), seed=1))
blk = pytorchcv_amd.set_compute_dtype(blk.to(dev), "bf16")
x = engine.NHWC(torch.randn(256, 56, 56, 64, device=dev).to(torch.bfloat16), 256, 56, 56, 64)
tune("d3c", 1)
with torch.no_grad():
    for _ in range(20): blk(x)
torch.cuda.synchronize()
d = dbg.cpu().view(-1, 4).to(torch.int64) & 0xFFFFFFFF
d = d[d[:, 3] > 0].float()
for i, name in enumerate(("K loop", "epilogue", "wait + barrier")):
    print("%-16s median %7.0f cycles   min %7.0f   max %7.0f" % (name, float(d[:, i].median()), float(d[:, i].min()), float(d[:, i].max())))
print("tile total median %.0f cycles (%d wave records)" % (float(d[:, :3].sum(1).median()), len(d)))
tune("dbg_lo", 0); tune("dbg_hi", 0); tune("d3c", -1)
