"""Dev tool: where does d3k differ from the generic kernel? (per channel block / per pixel block / per image row histogram)"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, os.path.join(R, "tests", "golden"))
import torch, util, pytorchcv_amd
from pytorchcv_amd import engine
from pytorchcv_amd.models.common.conv import conv3x3_block
N, Co, H = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (2, 128, 28)
DTY = sys.argv[4] if len(sys.argv) > 4 else "bf16"
C, W = 128, 28
dev = torch.device("cuda", 0)
blk = conv3x3_block(in_channels=C, out_channels=Co).eval()
blk.load_state_dict(util.synth_state_dict(blk.state_dict(), seed=79))
blk = pytorchcv_amd.set_compute_dtype(blk.to(dev), DTY)
x = util.synth_input(N, C, H, W, seed=25)
with torch.no_grad():
    xh = engine.from_nchw(x.to(dev), DTY, stem=False)
    with util.tuning(d3k=1): a = blk(xh).t.clone()
    with util.tuning(d3x3=0): b = blk(xh).t.clone()
a = a.reshape(N, H, W, Co).float().cpu(); b = b.reshape(N, H, W, Co).float().cpu()
bad = a != b
print("mismatches", int(bad.sum()), "of", bad.numel())
print("per channel (groups of 16):", [int(v) for v in bad.sum((0, 1, 2)).view(-1, 16).sum(1)])
print("per image row:", [int(v) for v in bad.sum((0, 2, 3))])
print("per column:", [int(v) for v in bad.sum((0, 1, 3))])
for n, h, w, c in bad.nonzero()[:8].tolist(): print("  n %d y %d x %d ch %d: %g vs %g" % (n, h, w, c, a[n, h, w, c], b[n, h, w, c]))
