import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/tests/golden")
import pytorchcv_amd, util
from pytorchcv_amd import engine
from pytorchcv_amd.models.common.conv import conv3x3_block
dev = torch.device("cuda", 0)
blk = conv3x3_block(in_channels=3, out_channels=32, stride=2).eval()
blk.load_state_dict(util.synth_state_dict(blk.state_dict(), seed=4))
x = util.synth_input(8, 3, 224, 224, seed=1).to(dev).repeat(64, 1, 1, 1).contiguous()
for dtype in ("fp16", "bf16"):
    blk = pytorchcv_amd.set_compute_dtype(blk.to(dev), dtype)
    for sw in (0, 1, 0, 1):
        with util.tuning(stem32=sw), torch.no_grad():
            for _ in range(3): blk(engine.network_input(x, dtype))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): blk(engine.network_input(x, dtype))
            e1.record(); torch.cuda.synchronize()
            print(dtype, "stem32 =", sw, "%.1f us" % (e0.elapsed_time(e1) * 100))
