"""Dev tool (diagnostic build: make -C pytorchcv_amd/csrc EXTRA=-DD3Q_CYCLES): shader cycles and real time of one block's K loop."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, pytorchcv_amd
from pytorchcv_amd import engine, _lib
from pytorchcv_amd.models.common.conv import conv3x3_block
from pytorchcv_amd.synth import synth_state_dict
dev = torch.device("cuda", 0); ctx = _lib.ctx_for(0)
dbg = torch.zeros(512, dtype=torch.int32, device=dev)
def tune(k, v): _lib.check(_lib.lib().pcv_set_tuning(ctx, k.encode(), ctypes.c_int(v).value), ctx)
ptr = dbg.data_ptr(); tune("dbg_lo", ctypes.c_int32(ptr & 0xFFFFFFFF).value); tune("dbg_hi", ctypes.c_int32(ptr >> 32).value)
for C, H in ((256, 14), (128, 28), (512, 7), (64, 56)):
    blk = conv3x3_block(in_channels=C, out_channels=C).eval()
    blk.load_state_dict(synth_state_dict(blk.state_dict(), seed=1))
    blk = pytorchcv_amd.set_compute_dtype(blk.to(dev), "bf16")
    x = engine.NHWC(torch.randn(256, H, H, C, device=dev).to(torch.bfloat16), 256, H, H, C)
    with torch.no_grad():
        for _ in range(30): blk(x)
    torch.cuda.synchronize()
    c, rt, kt = [int(v) & 0xFFFFFFFF for v in dbg[:3].cpu().tolist()]
    print("C=%d H=%d: K loop of one block: %d shader cycles, %d x 10 ns real time -> clock %.2f GHz; %d K-steps -> %.0f cycles per K-step" % (
        C, H, c, rt, c / (rt * 10.0), kt, c / kt))
tune("dbg_lo", 0); tune("dbg_hi", 0)
