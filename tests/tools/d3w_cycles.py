"""Dev tool (diagnostic build: make -C pytorchcv_amd/csrc EXTRA="-DD3W_CYCLES -DPCV_DBG_FLAGS"): phase stamps of every block of d3w_kernel -
entry -> prologue barrier -> start of the last epilogue -> exit (s_memrealtime, 10 ns units, and shader cycles)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, pytorchcv_amd
from pytorchcv_amd import engine, _lib
from pytorchcv_amd.models.common.conv import conv3x3_block
from pytorchcv_amd.synth import synth_state_dict
dev = torch.device("cuda", 0); ctx = _lib.ctx_for(0)
dbg = torch.zeros(256 * 2 * 8, dtype=torch.int32, device=dev)
def tune(k, v): _lib.check(_lib.lib().pcv_set_tuning(ctx, k.encode(), ctypes.c_int(v).value), ctx)
ptr = dbg.data_ptr(); tune("dbg_lo", ctypes.c_int32(ptr & 0xFFFFFFFF).value); tune("dbg_hi", ctypes.c_int32(ptr >> 32).value)
CASES = ((256, 14, 1, 0), (256, 14, 1, 32), (256, 14, 1, 64), (256, 14, 1, 96), (256, 14, 1, 128), (256, 14, 1, 256), (256, 14, 1, 32 + 64 + 128), (256, 14, 1, 32 + 64 + 256),
         (256, 14, 6, 0), (128, 28, 7, 0), (128, 28, 7, 32), (128, 28, 7, 64), (128, 28, 7, 96), (64, 56, 5, 0), (64, 56, 5, 32), (64, 56, 5, 96))
for C, H, shape, flags in CASES:
    blk = conv3x3_block(in_channels=C, out_channels=C).eval()
    blk.load_state_dict(synth_state_dict(blk.state_dict(), seed=1))
    blk = pytorchcv_amd.set_compute_dtype(blk.to(dev), "bf16")
    x = engine.NHWC(torch.randn(256, H, H, C, device=dev).to(torch.bfloat16), 256, H, H, C)
    tune("d3w", shape); tune("dbg", flags)
    with torch.no_grad():
        for _ in range(20): blk(x)
    torch.cuda.synchronize()
    dbg.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.no_grad():
        e0.record(); blk(x); e1.record()
    torch.cuda.synchronize()
    d = dbg.cpu().view(-1, 8).to(torch.int64) & 0xFFFFFFFF
    d = d[d[:, 7] > 0]
    t0 = d[:, 0]; start = (t0 - t0.min()) & 0xFFFFFFFF
    pro, loop_end, end = d[:, 1], d[:, 2], d[:, 3]
    ks = d[:, 7].float()
    clk = (d[:, 6].float() / end.float() / 10.0)       # cycles per 10 ns -> GHz x 100 ... cycles / (rt * 10 ns)
    print("C=%d H=%d shape %d dbg %d: event time %.1f us; %d wave records, K-steps per block %d..%d" % (C, H, shape, flags, e0.elapsed_time(e1) * 1e3, len(d), int(ks.min()), int(ks.max())))
    def stat(name, v): print("   %-34s median %7.2f us   min %7.2f   max %7.2f" % (name, float(v.float().median()) / 100, float(v.min()) / 100, float(v.max()) / 100))
    stat("entry -> prologue barrier", pro)
    stat("prologue barrier -> last epilogue", loop_end - pro)
    stat("last epilogue -> stores drained", end - loop_end)
    stat("entry -> exit", end)
    print("   clock %.2f GHz; K loop %.0f cycles per K-step (median block)" % (float(clk.median()) * 100 / 100, float(((d[:, 5] - d[:, 4]).float() / ks).median())))
tune("dbg_lo", 0); tune("dbg_hi", 0); tune("d3w", -1); tune("dbg", 0)
