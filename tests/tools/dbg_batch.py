"""Dev tool: find the first unit whose full-batch output differs from the 4-image forward / differs between repeats."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [R, os.path.join(R, "tests"), os.path.join(R, "tests", "golden")]
import torch, util
import pytorchcv_amd
from pytorchcv_amd import engine
from pytorchcv_amd.model_provider import get_model
name, batch = (sys.argv[1], int(sys.argv[2])) if len(sys.argv) > 2 else ("mobilenetv2_w1", 512)
dev = torch.device("cuda", 0)
logits, ids = util.model_golden(name)
net = get_model(name).eval()
net.load_state_dict(util.model_state(name, net.state_dict()), strict=True)
net = pytorchcv_amd.set_compute_dtype(net.to(dev), "bf16")
x4 = util.images(ids).to(dev)
x = x4.repeat(batch // 4, 1, 1, 1).contiguous()

def units(net):
    for n, m in net.features.named_children():
        if n.startswith("stage"):
            for un, u in m.named_children():
                yield n + "." + un, u
        else:
            yield n, m

def trace(xin):
    outs = []
    a = engine.from_nchw(xin, engine.compute_dtype_of(net), stem=True)
    for n, u in units(net):
        a = u(a)
        outs.append((n, a.t.clone()))
    return outs

for sw in ({}, {"head": 0}, {"mbw": 0}):
    with torch.no_grad(), util.tuning(**sw):
        ref4 = trace(x4)
        for rep in range(6):
            full = trace(x)
            torch.cuda.synchronize()
            for (n, t4), (_, t) in zip(ref4, full):
                want = t4.repeat(batch // 4, 1, 1, 1)
                if not torch.equal(t, want):
                    bad = (t != want)
                    idx = bad.nonzero()
                    print(sw, "rep", rep, "first differing unit:", n, tuple(t.shape), "elements", int(bad.sum()), "images",
                          sorted(set(idx[:, 0].tolist()))[:12], "rows", sorted(set(idx[:, 1].tolist()))[:8], "cols", sorted(set(idx[:, 2].tolist()))[:8],
                          "chans", sorted(set(idx[:, 3].tolist()))[:8], flush=True)
                    break
            else:
                print(sw, "rep", rep, "all units identical", flush=True)
