"""Dev tool: in-process A/B of kernel variants on single conv layers (interleaved rounds, median of HIP-event times).
Usage: python tests/tools/bench_conv.py  (edit LAYERS / VARIANTS below)"""
import sys, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import pytorchcv_amd
from pytorchcv_amd import engine, _lib
from pytorchcv_amd.models.common.conv import ConvBlock
from pytorchcv_amd.synth import synth_state_dict

dev = torch.device("cuda", 0)
LAYERS = [  # (N, Cin, Cout, H, k, stride, groups, residual)
    (256, 64, 64, 56, 3, 1, 1, False), (256, 128, 128, 28, 3, 1, 1, False), (256, 256, 256, 14, 3, 1, 1, False),
    (256, 512, 512, 7, 3, 1, 1, False),
]
SHAPES = ["256x112", "128x224", "64x448", "256x64", "128x128", "64x256", "256x112k1", "64x448k1"]     # d3q_inst.hpp order
VARIANTS = {"generic": {"d3x3": 0}, "auto": {"d3x3": -1}}
VARIANTS.update({"d3:" + n: {"d3x3": i + 1} for i, n in enumerate(SHAPES)})
if len(sys.argv) > 1:
    exec(open(sys.argv[1]).read())       # a file may redefine LAYERS / VARIANTS

ctx = _lib.ctx_for(0)
def tune(d):
    for k, v in d.items():
        _lib.check(_lib.lib().pcv_set_tuning(ctx, k.encode(), int(v)), ctx)

for (N, C, Co, H, k, s, g, res) in LAYERS:
    blk = ConvBlock(C, Co, k, stride=s, padding=k // 2, groups=g).eval()
    blk.load_state_dict(synth_state_dict(blk.state_dict(), seed=1))
    blk = pytorchcv_amd.set_compute_dtype(blk.to(dev), "bf16")
    x = engine.NHWC(torch.randn(N, H, H, C, device=dev).to(torch.bfloat16), N, H, H, C)
    Ho = (H + 2 * (k // 2) - k) // s + 1
    r = engine.NHWC(torch.randn(N, Ho, Ho, Co, device=dev).to(torch.bfloat16), N, Ho, Ho, Co) if res else None
    times = {v: [] for v in VARIANTS}
    if os.environ.get("BENCH_GRAPH"):
        # BENCH_GRAPH=1: 20 launches per variant captured into a hipGraph and replayed - the figure for launches under ~40 us, where the
        # eager loop below measures the host's launch rate (~28 us per call) instead of the kernel
        with torch.no_grad():
            graphs = {}
            for v, t in VARIANTS.items():
                tune(t); blk(x, residual=r); torch.cuda.synchronize()
                cg = torch.cuda.CUDAGraph()
                with torch.cuda.graph(cg):
                    for _ in range(20):
                        y = blk(x, residual=r)
                graphs[v] = (cg, y)
            for rnd in range(5):
                for v, (cg, _) in graphs.items():
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(); cg.replay(); e1.record(); torch.cuda.synchronize()
                    times[v].append(e0.elapsed_time(e1) / 20 * 1e3)
        Ho_ = (H + 2 * (k // 2) - k) // s + 1
        flops = 2.0 * N * Ho_ * Ho_ * Co * (C // g) * k * k
        print("N%d %dx%d C%d->%d k%d s%d g%d%s (graph replay):" % (N, H, H, C, Co, k, s, g, " +res" if res else ""), flush=True)
        for v, t in times.items():
            print("    %-12s %7.1f us  %6.0f TF   (min %.1f)" % (v, statistics.median(t), flops / statistics.median(t) / 1e6, min(t)), flush=True)
        continue
    with torch.no_grad():
        for v, t in VARIANTS.items():
            tune(t); blk(x, residual=r)
        torch.cuda.synchronize()
        for rnd in range(5):
            for v, t in VARIANTS.items():
                tune(t)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    blk(x, residual=r)
                e1.record(); torch.cuda.synchronize()
                times[v].append(e0.elapsed_time(e1) / 5 * 1e3)
    flops = 2.0 * N * Ho * Ho * Co * (C // g) * k * k
    print("N%d %dx%d C%d->%d k%d s%d g%d%s:" % (N, H, H, C, Co, k, s, g, " +res" if res else ""), flush=True)
    for v, t in times.items():
        print("    %-12s %7.1f us  %6.0f TF   (min %.1f)" % (v, statistics.median(t), flops / statistics.median(t) / 1e6, min(t)), flush=True)
tune({"d3x3": -1, "d3w": -1, "d3c": -1, "d3k": -1, "d3i": -1, "d1i": -1, "p1r": -1, "d1x1": -1, "dbg": 0, "dw_flags": 0, "dw_th": 0})
