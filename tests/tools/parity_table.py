"""Dev tool: the parity table of DESIGN.md section 3 - max |logit difference| of every fixture net against the reference golden."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [R, os.path.join(R, "tests"), os.path.join(R, "tests", "golden")]
import torch, util
import pytorchcv_amd
from pytorchcv_amd.model_provider import get_model
dev = torch.device("cuda", 0)
print("| model | default mode | fp32 | bf16 | fp16 | top-1 (default mode) |")
for name in util.MODELS:
    logits, ids = util.model_golden(name)
    x = util.images(ids).to(dev)
    row = []
    top = True
    mode = None
    for dt in ("auto", "fp32", "bf16", "fp16"):
        net = get_model(name).eval()
        net.load_state_dict(util.model_state(name, net.state_dict()), strict=True)
        net = pytorchcv_amd.set_compute_dtype(net.to(dev), dt)
        with torch.no_grad():
            y = net(x).float().cpu()
        row.append(float((y - logits).abs().max()))
        if dt == "auto":
            top = bool(torch.equal(y.argmax(1), logits.argmax(1)))
            mode = pytorchcv_amd.engine.compute_dtype_of(net)
        del net
    print("| %s | %s %.1e | %.1e | %.1e | %.1e | %s |" % (name, mode, row[0], row[1], row[2], row[3], "identical" if top else "DIFFERS"), flush=True)
