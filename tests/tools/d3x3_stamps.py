"""Dev tool (diagnostic build only: `make -C pytorchcv_amd/csrc clean all EXTRA=-DD3X3_STAMPS`): where a steady-state stage of
d3x3_kernel spends its cycles. Prints, per wave of one block, the median over 7 stages of the section times between the stamps
of d3x3_conv.hpp (work of interval i, then the wait at the barrier that ends it; for interval 3 also the vmcnt wait).
Usage: python tests/tools/d3x3_stamps.py <C> <H> <shape index 1..8> [N]"""
import sys, os, statistics, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import pytorchcv_amd
from pytorchcv_amd import engine, _lib
from pytorchcv_amd.models.common.conv import conv3x3_block
from pytorchcv_amd.synth import synth_state_dict

C, H, shape = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
N = int(sys.argv[4]) if len(sys.argv) > 4 else 256
dev = torch.device("cuda", 0)
ctx = _lib.ctx_for(0)
dbg = torch.zeros(8 * 64, dtype=torch.int32, device=dev)
def tune(k, v):
    _lib.check(_lib.lib().pcv_set_tuning(ctx, k.encode(), ctypes.c_int(v).value), ctx)
ptr = dbg.data_ptr()
tune("dbg_lo", ctypes.c_int32(ptr & 0xFFFFFFFF).value); tune("dbg_hi", ctypes.c_int32(ptr >> 32).value); tune("d3x3", shape)
blk = conv3x3_block(in_channels=C, out_channels=C).eval()
blk.load_state_dict(synth_state_dict(blk.state_dict(), seed=1))
blk = pytorchcv_amd.set_compute_dtype(blk.to(dev), "bf16")
x = engine.NHWC(torch.randn(N, H, H, C, device=dev).to(torch.bfloat16), N, H, H, C)
with torch.no_grad():
    for _ in range(5):
        blk(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        blk(x)
    e1.record()
torch.cuda.synchronize()
print("C=%d H=%d shape=%d N=%d: %.1f us per launch in this (stamped) build" % (C, H, shape, N, e0.elapsed_time(e1) / 20 * 1e3))
st = dbg.cpu().numpy().astype("uint32").reshape(8, 64)
ks1 = ["I0 work", "I0 barrier", "I1 work", "I1 barrier", "I2 work", "I2 barrier", "I3 work", "vmcnt wait", "last barrier"]
print("cycles per section (median of 3 samples, one section timed per stage); KS = 2 shapes have no I1 barrier .. I2 barrier")
print("wave " + " ".join("%12s" % n for n in ks1) + "   sum")
for w in range(8):
    d = {}
    for q in range(27):
        d.setdefault(q % 9, []).append((int(st[w, 2 * q + 1]) - int(st[w, 2 * q])) & 0xFFFFFFFF)
    med = [statistics.median(d[i]) for i in range(9)]
    print("%4d " % w + " ".join("%12d" % m for m in med) + "   %6d" % sum(med))
tune("dbg_lo", 0); tune("dbg_hi", 0); tune("d3x3", -1)
