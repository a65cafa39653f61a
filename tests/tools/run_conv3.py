"""Dev tool: run ONE dense 3x3 layer a few times with a forced kernel (for rocprofv3 --pmc / --kernel-trace runs).
   python tests/tools/run_conv3.py <C> <H> <key=value,...> [iters]     e.g.  64 56 d3c=1"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, pytorchcv_amd
from pytorchcv_amd import engine, _lib
from pytorchcv_amd.models.common.conv import conv3x3_block
from pytorchcv_amd.synth import synth_state_dict
C, H = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda", 0); ctx = _lib.ctx_for(0)
for kv in filter(None, (sys.argv[3] if len(sys.argv) > 3 else "").split(",")):
    k, v = kv.split("=")
    _lib.check(_lib.lib().pcv_set_tuning(ctx, k.encode(), int(v)), ctx)
blk = conv3x3_block(in_channels=C, out_channels=C).eval()
blk.load_state_dict(synth_state_dict(blk.state_dict(), seed=1))
blk = pytorchcv_amd.set_compute_dtype(blk.to(dev), "bf16")
x = engine.NHWC(torch.randn(256, H, H, C, device=dev).to(torch.bfloat16), 256, H, H, C)
with torch.no_grad():
    for _ in range(int(sys.argv[4]) if len(sys.argv) > 4 else 5): blk(x)
torch.cuda.synchronize()
print("done")
