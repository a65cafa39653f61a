"""Dev tool: measured fp32 distance to the oracle of the registry models test_gpu_registry.py checks (to set its bounds)."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [R, os.path.join(R, "tests"), os.path.join(R, "tests", "golden")]
import torch, util
import pytorchcv_amd
from oracle import refnet
from pytorchcv_amd.model_provider import get_model
import test_gpu_registry as T
dev = torch.device("cuda", 0)
for name, size, check in T._NETS:
    if not check:
        continue
    net = get_model(name).eval()
    sd = util.synth_state_dict(net.state_dict(), seed=5)
    net.load_state_dict(sd, strict=True)
    x = util.synth_input(2, 3, size, size, seed=9)
    net = pytorchcv_amd.set_compute_dtype(net.to(dev), "fp32")
    with torch.no_grad():
        y = net(x.to(dev)).cpu()
    ref = refnet.forward(name, {k: v.float() for k, v in sd.items()}, x)
    ref64 = refnet.forward(name, {k: v.double() for k, v in sd.items()}, x.double()) if os.environ.get("F64") else None
    m = max(1.0, float(ref.abs().max()))
    print("%-28s max|ref| %9.3g  rel err %.2e%s" % (name, float(ref.abs().max()), float((y - ref).abs().max()) / m,
          "  oracle fp32 vs fp64 %.2e  gpu vs fp64 %.2e" % (float((ref.double() - ref64).abs().max()) / m, float((y.double() - ref64).abs().max()) / m) if ref64 is not None else ""), flush=True)
