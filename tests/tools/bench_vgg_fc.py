import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, pytorchcv_amd
from pytorchcv_amd import engine, _lib
from pytorchcv_amd.models.common.conv import ConvBlock
from pytorchcv_amd.synth import synth_state_dict
dev = torch.device("cuda", 0); ctx = _lib.ctx_for(0)
def tune(d):
    for k, v in d.items(): _lib.check(_lib.lib().pcv_set_tuning(ctx, k.encode(), int(v)), ctx)
for (C, Co, H, k) in ((512, 4096, 7, 7), (4096, 4096, 1, 1), (4096, 1000, 1, 1)):
    blk = ConvBlock(C, Co, k, stride=1, padding=0).eval()
    blk.load_state_dict(synth_state_dict(blk.state_dict(), seed=1))
    blk = pytorchcv_amd.set_compute_dtype(blk.to(dev), "bf16")
    x = engine.NHWC(torch.randn(128, H, H, C, device=dev).to(torch.bfloat16), 128, H, H, C)
    for name, t in (("auto", {"tile": -1}), ("32x256", {"tile": 0}), ("64x256", {"tile": 1}), ("128x128", {"tile": 2}), ("256x64", {"tile": 3}), ("64x128", {"tile": 4}), ("128x64", {"tile": 5}), ("32x256np", {"tile": 0, "persist": 0})):
        try:
            tune({"tile": -1, "persist": 1}); tune(t)
            with torch.no_grad():
                for _ in range(3): blk(x)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10): blk(x)
                e1.record(); torch.cuda.synchronize()
            print("%d->%d k%d  %-9s %7.1f us" % (C, Co, k, name, e0.elapsed_time(e1) * 100))
        except Exception as ex:
            print(name, "ERR", str(ex)[:80])
tune({"tile": -1, "persist": 1})
