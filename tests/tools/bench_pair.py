"""Dev tool: fused 1x1 pair (pcv_conv1x1_pair_fused) against the two separate launches, in one process."""
import sys, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn as nn
import pytorchcv_amd
from pytorchcv_amd import engine
from pytorchcv_amd.models.common.conv import conv1x1_block, conv_block_pair
from pytorchcv_amd.synth import synth_state_dict

dev = torch.device("cuda", 0)
N, H = 256, 56
first = conv1x1_block(in_channels=64, out_channels=256, activation=None).eval()
second = conv1x1_block(in_channels=256, out_channels=64).eval()
first.load_state_dict(synth_state_dict(first.state_dict(), seed=1))
second.load_state_dict(synth_state_dict(second.state_dict(), seed=2))
first = pytorchcv_amd.set_compute_dtype(first.to(dev), "bf16")
second = pytorchcv_amd.set_compute_dtype(second.to(dev), "bf16")
x = engine.NHWC(torch.randn(N, H, H, 64, device=dev).to(torch.bfloat16), N, H, H, 64)
r = engine.NHWC(torch.randn(N, H, H, 256, device=dev).to(torch.bfloat16), N, H, H, 256)
relu = nn.ReLU()
def sep():
    return second(first(x, residual=r, post_act=relu))
def only1():
    return first(x, residual=r, post_act=relu)
def fused():
    return conv_block_pair(first, x, r, relu, second)
from pytorchcv_amd import _lib
ctx = _lib.ctx_for(0)
def tune(v):
    _lib.check(_lib.lib().pcv_set_tuning(ctx, b"pair_pb", v), ctx)
def fused4():
    tune(4); return fused()
def fused2():
    tune(2); return fused()
times = {"conv3+res": [], "separate": [], "fused pb4": [], "fused pb2": []}
fns = {"conv3+res": only1, "separate": sep, "fused pb4": fused4, "fused pb2": fused2}
with torch.no_grad():
    for f in fns.values():
        f()
    torch.cuda.synchronize()
    for rnd in range(7):
        for k, f in fns.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                f()
            e1.record(); torch.cuda.synchronize()
            times[k].append(e0.elapsed_time(e1) / 5 * 1e3)
mb = N * H * H * (64 + 256 + 256 + 64) * 2 / 1e6
print("  ".join("%s %.1f us" % (k, statistics.median(t)) for k, t in times.items()), " (fused moves %.0f MB)" % mb)
