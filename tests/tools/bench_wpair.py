"""Dev tool: wide fused 1x1 pair (128 -> 512 -> 128, wpair1x1.hpp) against the two separate launches, in one process."""
import sys, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn as nn
import pytorchcv_amd
from pytorchcv_amd import engine
from pytorchcv_amd.models.common.conv import conv1x1_block, conv_block_pair
from pytorchcv_amd.synth import synth_state_dict

dev = torch.device("cuda", 0)
CM = int(sys.argv[1]) if len(sys.argv) > 1 else 128
N, H = 256, {64: 56, 128: 28, 256: 14}[CM]
first = conv1x1_block(in_channels=CM, out_channels=4 * CM, activation=None).eval()
second = conv1x1_block(in_channels=4 * CM, out_channels=CM).eval()
first.load_state_dict(synth_state_dict(first.state_dict(), seed=1))
second.load_state_dict(synth_state_dict(second.state_dict(), seed=2))
first = pytorchcv_amd.set_compute_dtype(first.to(dev), "bf16")
second = pytorchcv_amd.set_compute_dtype(second.to(dev), "bf16")
x = engine.NHWC(torch.randn(N, H, H, CM, device=dev).to(torch.bfloat16), N, H, H, CM)
r = engine.NHWC(torch.randn(N, H, H, 4 * CM, device=dev).to(torch.bfloat16), N, H, H, 4 * CM)
relu = nn.ReLU()
fns = {"conv3+res": lambda: first(x, residual=r, post_act=relu),
       "separate": lambda: second(first(x, residual=r, post_act=relu)),
       "fused": lambda: conv_block_pair(first, x, r, relu, second)}
times = {k: [] for k in fns}
with torch.no_grad():
    for f in fns.values():
        assert f() is not None
    torch.cuda.synchronize()
    for rnd in range(7):
        for k, f in fns.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                f()
            e1.record(); torch.cuda.synchronize()
            times[k].append(e0.elapsed_time(e1) / 5 * 1e3)
mb = N * H * H * (CM + 4 * CM + 4 * CM + CM) * 2 / 1e6
print("  ".join("%s %.1f us" % (k, statistics.median(t)) for k, t in times.items()), " (fused moves %.0f MB)" % mb)
