"""Dev tool (diagnostic build: make -C pytorchcv_amd/csrc EXTRA=-DP1R_CYCLES): shader cycles of one tile of p1r_kernel per wave -
the 64 steps with the interleaved epilogue (+ the last unit's), the wait for the next tile's pieces, the barrier.
Usage: python tests/tools/p1r_cycles.py [Cin Cout H stride residual]"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, pytorchcv_amd
from pytorchcv_amd import engine, _lib
from pytorchcv_amd.models.common.conv import conv1x1_block
from pytorchcv_amd.synth import synth_state_dict
dev = torch.device("cuda", 0); ctx = _lib.ctx_for(0)
C, Co, H, s, res = (int(v) for v in (sys.argv[1:6] if len(sys.argv) > 5 else (256, 512, 28, 1, 0)))
dbg = torch.zeros(256 * 8 * 4 + 256 * 32, dtype=torch.int32, device=dev)
def tune(k, v): _lib.check(_lib.lib().pcv_set_tuning(ctx, k.encode(), ctypes.c_int(v).value), ctx)
ptr = dbg.data_ptr(); tune("dbg_lo", ctypes.c_int32(ptr & 0xFFFFFFFF).value); tune("dbg_hi", ctypes.c_int32(ptr >> 32).value)
blk = conv1x1_block(in_channels=C, out_channels=Co, stride=s).eval()
blk.load_state_dict(synth_state_dict(blk.state_dict(), seed=1))
blk = pytorchcv_amd.set_compute_dtype(blk.to(dev), "bf16")
x = engine.NHWC(torch.randn(256, H, H, C, device=dev).to(torch.bfloat16), 256, H, H, C)
Ho = (H - 1) // s + 1
r = engine.NHWC(torch.randn(256, Ho, Ho, Co, device=dev).to(torch.bfloat16), 256, Ho, Ho, Co) if res else None
tune("p1r", 1)
with torch.no_grad():
    for _ in range(20): blk(x, residual=r, post_act=torch.nn.ReLU() if res else None)
torch.cuda.synchronize()
e = (dbg.cpu()[256 * 8 * 4:].view(256, 16, 2).to(torch.int64) & 0xFFFFFFFF).float()
d = dbg.cpu()[:256 * 8 * 4].view(-1, 4).to(torch.int64) & 0xFFFFFFFF
d = d[d[:, 3] > 0].float()
print("%d -> %d, %dx%d, stride %d%s" % (C, Co, H, H, s, " + residual" if res else ""))
for i, name in enumerate(("steps + epilogue", "wait for pieces", "barrier")):
    print("%-18s median %7.0f cycles   min %7.0f   max %7.0f" % (name, float(d[:, i].median()), float(d[:, i].min()), float(d[:, i].max())))
print("tile total median %.0f cycles (%d wave records); the tile's MFMAs: %d matrix-pipe cycles per SIMD" % (float(d[:, :3].sum(1).median()), len(d), 2 * 4096 if C == 256 else 2 * 2048))
print("per tile (wave 0 of every block): start after the kernel's first instruction / duration, medians over the blocks that ran it")
for tl in range(15):
    if tl >= 15: break
    m = e[:, tl, 1] > 0
    if int(m.sum()) == 0: break
    print("  tile %2d: %3d blocks   start %8.0f   duration %7.0f (min %7.0f max %7.0f)" % (tl, int(m.sum()), float(e[m, tl, 0].median()), float(e[m, tl, 1].median()), float(e[m, tl, 1].min()), float(e[m, tl, 1].max())))
pr = dbg.cpu()[256 * 8 * 4:].view(256, 32)[:, 30:32].to(torch.int64) & 0xFFFFFFFF
names = ("kernel start -> weights requested from", "issuing the weight loads", "waiting for them", "barrier")
vals = (pr[:, 0] & 0xFFFF, pr[:, 0] >> 16, pr[:, 1] & 0xFFFF, pr[:, 1] >> 16)
for n, v in zip(names, vals):
    print("  prologue, %-40s median %6.0f   min %6.0f   max %6.0f" % (n, float(v.float().median()), float(v.min()), float(v.max())))
last = (e[:, :15, 0] + e[:, :15, 1]).max()
print("last tile ends %.0f cycles after its block's start" % float(last))
tune("dbg_lo", 0); tune("dbg_hi", 0); tune("p1r", -1)
