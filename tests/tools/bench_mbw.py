"""Dev tool: fused inverted-residual units (pcv_mbconv_fused) against their three launches, in one process, interleaved.
Usage: python tests/tools/bench_mbw.py [dtype=fp16] [N=512] [unit ...]     unit = Cin:Cout:H:stride:expand (expand 0 = LinearBottleneck(expansion=False))
Default: MobileNetV2 x1.0's fused units at batch 512. Variants: three launches, mbw (register-resident kernel off), default routing."""
import sys, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "golden"))
import torch
import util
import pytorchcv_amd
from pytorchcv_amd import engine
from pytorchcv_amd.models.mobilenetv2 import LinearBottleneck
from pytorchcv_amd.models.common.conv import mbconv_chain
from pytorchcv_amd.models.common.activ import create_activation_layer
from pytorchcv_amd.synth import synth_state_dict
dev = torch.device("cuda", 0)
args = sys.argv[1:]
dtype = args.pop(0) if args and args[0] in ("bf16", "fp16") else "fp16"
N = int(args.pop(0)) if args and args[0].isdigit() else 512
V2 = ["32:16:112:1:0", "16:24:112:2:1", "24:24:56:1:1", "24:32:56:2:1", "32:32:28:1:1", "32:64:28:2:1", "64:64:14:1:1", "64:96:14:1:1", "96:96:14:1:1"]
units = args or V2
tdt = {"bf16": torch.bfloat16, "fp16": torch.float16}[dtype]
for spec in units:
    Cin, Cout, H, stride, expand = (int(v) for v in spec.split(":"))
    unit = LinearBottleneck(in_channels=Cin, out_channels=Cout, stride=stride, expansion=bool(expand), remove_exp_conv=False,
                            activation=(lambda: create_activation_layer("relu6"))).eval()
    unit.load_state_dict(synth_state_dict(unit.state_dict(), seed=3))
    unit = pytorchcv_amd.set_compute_dtype(unit.to(dev), dtype)
    x = engine.NHWC(torch.randn(N, H, H, Cin, device=dev).to(tdt), N, H, H, Cin)
    res = x if unit.residual else None
    Ho = (H - 1) // stride + 1
    mb = (N * H * H * Cin + N * Ho * Ho * Cout) * 2 / 1e6

    def fused():
        return mbconv_chain(unit.conv1, unit.conv2, unit.conv3, x, residual=res)

    def with_tuning(**kw):
        def f():
            with util.tuning(**kw):
                return fused()
        return f
    fns = {"three launches": lambda: unit.conv3(unit.conv2(unit.conv1(x)), residual=res), "mbw": with_tuning(mbr=0), "mbr_regs": with_tuning(mbr_xl=0), "default": fused}
    if os.environ.get("BENCH_MBW_ONLY"):
        fns = {k: f for k, f in fns.items() if k in os.environ["BENCH_MBW_ONLY"].split(",") or k == "three launches"}
    times = {k: [] for k in fns}
    with torch.no_grad():
        outs = {k: f() for k, f in fns.items()}
        torch.cuda.synchronize()
        ref = outs["three launches"].t.float()
        diffs = {k: (float((o.t.float() - ref).abs().max()) if o is not None else None) for k, o in outs.items()}
        for rnd in range(7):
            for k, f in fns.items():
                if outs[k] is None or (os.environ.get("BENCH_MBW_ONLY") and k == "three launches"):
                    continue
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    f()
                e1.record(); torch.cuda.synchronize()
                times[k].append(e0.elapsed_time(e1) / 5 * 1e3)
    print("%d->%d->%d @%dx%d s%d N%d %s (%.0f MB in+out): " % (Cin, unit.conv2.conv.weight.shape[0], Cout, H, H, stride, N, dtype, mb) +
          "  ".join("%s %.1f us (%.2f TB/s, maxdiff %.3g)" % (k, statistics.median(t), mb / statistics.median(t), diffs[k]) for k, t in times.items() if t),
          flush=True)
