"""Dev tool: run ONE 3x3 layer a few times with a tuning variant (for rocprofv3 --pmc attribution).
   python tests/tools/one_conv.py <C> <H> <conv3 0|1> [c3flags]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import pytorchcv_amd
from pytorchcv_amd import engine, _lib
from pytorchcv_amd.models.common.conv import ConvBlock
from pytorchcv_amd.synth import synth_state_dict
C, H, conv3 = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
flags = int(sys.argv[4]) if len(sys.argv) > 4 else 0
dev = torch.device("cuda", 0)
ctx = _lib.ctx_for(0)
for k, v in (("conv3", conv3), ("c3flags", flags)):
    _lib.check(_lib.lib().pcv_set_tuning(ctx, k.encode(), v), ctx)
blk = ConvBlock(C, C, 3, padding=1).eval()
blk.load_state_dict(synth_state_dict(blk.state_dict(), seed=1))
blk = pytorchcv_amd.set_compute_dtype(blk.to(dev), "bf16")
x = engine.NHWC(torch.randn(256, H, H, C, device=dev).to(torch.bfloat16), 256, H, H, C)
with torch.no_grad():
    for _ in range(6):
        blk(x)
torch.cuda.synchronize()
