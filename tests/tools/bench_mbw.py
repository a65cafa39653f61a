"""Dev tool: one fused inverted-residual unit (pcv_mbconv_fused) against its three launches, in one process.
Usage: python tests/tools/bench_mbw.py <Cin> <Cout> <H> [stride] [N]"""
import sys, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import pytorchcv_amd
from pytorchcv_amd import engine
from pytorchcv_amd.models.mobilenetv2 import LinearBottleneck
from pytorchcv_amd.models.common.conv import mbconv_chain
from pytorchcv_amd.synth import synth_state_dict
dev = torch.device("cuda", 0)
Cin, Cout, H = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
stride = int(sys.argv[4]) if len(sys.argv) > 4 else 1
N = int(sys.argv[5]) if len(sys.argv) > 5 else 512
from pytorchcv_amd.models.common.activ import create_activation_layer
unit = LinearBottleneck(in_channels=Cin, out_channels=Cout, stride=stride, expansion=True, remove_exp_conv=True,
                        activation=(lambda: create_activation_layer("relu6"))).eval()
unit.load_state_dict(synth_state_dict(unit.state_dict(), seed=3))
unit = pytorchcv_amd.set_compute_dtype(unit.to(dev), "bf16")
x = engine.NHWC(torch.randn(N, H, H, Cin, device=dev).to(torch.bfloat16), N, H, H, Cin)
res = x if unit.residual else None
fns = {"three launches": lambda: unit.conv3(unit.conv2(unit.conv1(x)), residual=res),
       "fused": lambda: mbconv_chain(unit.conv1, unit.conv2, unit.conv3, x, residual=res)}
times = {k: [] for k in fns}
with torch.no_grad():
    for k, f in fns.items():
        assert f() is not None, k
    torch.cuda.synchronize()
    for rnd in range(7):
        for k, f in fns.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                f()
            e1.record(); torch.cuda.synchronize()
            times[k].append(e0.elapsed_time(e1) / 5 * 1e3)
print("%d->%d->%d @%dx%d s%d N%d: " % (Cin, 6 * Cin, Cout, H, H, stride, N) + "  ".join("%s %.1f us" % (k, statistics.median(t)) for k, t in times.items()))
