"""Dev tool: where does p1r differ from the generic kernel? (channel / pixel histogram of mismatches)"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, os.path.join(R, "tests", "golden"))
import torch, util, pytorchcv_amd
from pytorchcv_amd import engine
from pytorchcv_amd.models.common.conv import conv1x1_block
N, C, Co, H, W, res, s = (int(v) for v in sys.argv[1:8]) if len(sys.argv) > 7 else (4, 512, 1024, 14, 14, 1, 1)
mode = int(sys.argv[8]) if len(sys.argv) > 8 else 1
dev = torch.device("cuda", 0)
blk = conv1x1_block(in_channels=C, out_channels=Co, stride=s).eval()
blk.load_state_dict(util.synth_state_dict(blk.state_dict(), seed=85))
blk = pytorchcv_amd.set_compute_dtype(blk.to(dev), "bf16")
x = util.synth_input(N, C, H, W, seed=31)
r = util.synth_input(N, Co, (H - 1) // s + 1, (W - 1) // s + 1, seed=32) if res else None
with torch.no_grad():
    xh = engine.from_nchw(x.to(dev), "bf16", stem=False)
    rh = engine.from_nchw(r.to(dev), "bf16", stem=False) if res else None
    with util.tuning(p1r=mode):
        a = blk(xh, residual=rh, post_act=torch.nn.ReLU() if res else None).t.clone()
    with util.tuning(d1x1=0):
        b = blk(xh, residual=rh, post_act=torch.nn.ReLU() if res else None).t.clone()
a = a.reshape(-1, Co).float().cpu(); b = b.reshape(-1, Co).float().cpu()
bad = a != b
print("mismatches", int(bad.sum()), "of", bad.numel())
pc = bad.sum(0); pp = bad.sum(1)
print("per channel (groups of 16):", [int(v) for v in pc.view(-1, 16).sum(1)])
print("per pixel (groups of 16):", [int(v) for v in pp.view(-1, 16).sum(1)][:60])
i = bad.nonzero()[:8]
for m, c in i.tolist(): print("  px %d ch %d: %g vs %g" % (m, c, a[m, c], b[m, c]))
