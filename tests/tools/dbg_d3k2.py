"""Dev tool: d3k against the generic kernel with a one-tap filter (tap r, q; channel c -> output channel c): which input pixel does each output see?"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, os.path.join(R, "tests", "golden"))
import torch, util, pytorchcv_amd
from pytorchcv_amd import engine
from pytorchcv_amd.models.common.conv import conv3x3_block
N, H, C, W = 1, 8, 128, 28
dev = torch.device("cuda", 0)
for (r, q, ci) in ((2, 1, 0), (2, 0, 0), (2, 2, 0), (2, 1, 64), (1, 1, 0), (0, 1, 70)):
    blk = conv3x3_block(in_channels=C, out_channels=C).eval()
    sd = blk.state_dict()
    for k in sd:
        if k.endswith("conv.weight"):
            w = torch.zeros_like(sd[k]); w[0, ci, r, q] = 1.0; sd[k] = w
        elif k.endswith("bn.weight"): sd[k] = torch.ones_like(sd[k])
        elif k.endswith("bn.bias"): sd[k] = torch.zeros_like(sd[k])
        elif k.endswith("running_mean"): sd[k] = torch.zeros_like(sd[k])
        elif k.endswith("running_var"): sd[k] = torch.ones_like(sd[k])
    blk.load_state_dict(sd)
    blk = pytorchcv_amd.set_compute_dtype(blk.to(dev), "bf16")
    x = torch.zeros(N, C, H, W)
    for y in range(H):
        for xx in range(W): x[0, ci, y, xx] = 1 + y * 32 + xx          # value encodes the position (exact in bf16 up to 256)
    with torch.no_grad():
        xh = engine.from_nchw(x.to(dev), "bf16", stem=False)
        with util.tuning(d3k=1): a = blk(xh).t.clone()
        with util.tuning(d3x3=0): b = blk(xh).t.clone()
    a = a.reshape(N, H, W, C)[0, :, :, 0].float().cpu(); b = b.reshape(N, H, W, C)[0, :, :, 0].float().cpu()
    print("tap r=%d q=%d channel %d: mismatching outputs %d" % (r, q, ci, int((a != b).sum())))
    for y in range(H):
        if bool((a[y] != b[y]).any()):
            print("  row %d d3k    :" % y, [int(v) for v in a[y]])
            print("  row %d generic:" % y, [int(v) for v in b[y]])
