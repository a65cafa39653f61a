"""Dev experiment: does running a stage of bottleneck units in image chunks (intermediates resident in the 256 MiB
Infinity Cache) beat one pass over the whole batch?  python tests/tools/exp_chunk_stage.py"""
import sys, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import pytorchcv_amd
from pytorchcv_amd import engine
from pytorchcv_amd.models.resnet import ResUnit
from pytorchcv_amd.synth import synth_state_dict

dev = torch.device("cuda", 0)
for (C, H, nunits) in ((256, 56, 3), (512, 28, 4), (1024, 14, 6), (2048, 7, 3)):
    units = []
    for i in range(nunits):
        u = ResUnit(in_channels=C, out_channels=C, stride=1, bottleneck=True, conv1_stride=True).eval()
        u.load_state_dict(synth_state_dict(u.state_dict(), seed=i))
        units.append(pytorchcv_amd.set_compute_dtype(u.to(dev), "bf16"))
    N = 256
    xt = torch.randn(N, H, H, C, device=dev).to(torch.bfloat16)

    def run(chunk):
        outs = []
        for n0 in range(0, N, chunk):
            a = engine.NHWC(xt[n0:n0 + chunk], chunk, H, H, C)
            for u in units:
                a = u(a)
            outs.append(a)
        return outs

    res = {}
    with torch.no_grad():
        for chunk in (256, 128, 64, 32):
            run(chunk)
        torch.cuda.synchronize()
        for rnd in range(5):
            for chunk in (256, 128, 64, 32):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    run(chunk)
                e1.record(); torch.cuda.synchronize()
                res.setdefault(chunk, []).append(e0.elapsed_time(e1) / 3 * 1e3)
    print("C%d %dx%d x%d units:" % (C, H, H, nunits), "  ".join("chunk %d: %.0f us" % (c, statistics.median(t)) for c, t in res.items()), flush=True)
