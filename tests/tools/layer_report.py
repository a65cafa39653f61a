"""Dev tool: run one forward of a workload with every conv launch bracketed by events; print per-launch us, TB/s, TFLOP/s."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, json
import pytorchcv_amd
from pytorchcv_amd import engine
from pytorchcv_amd.model_provider import get_model
from pytorchcv_amd.synth import synth_state_dict, synth_input
name, batch = (sys.argv[1], int(sys.argv[2])) if len(sys.argv) > 2 else ("resnet50", 256)
dev = torch.device("cuda", 0)
net = get_model(name).eval()
cal = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "golden", "calib_%s.json" % name)
calib = {k: tuple(v) for k, v in json.load(open(cal)).items()} if os.path.exists(cal) else None
net.load_state_dict(synth_state_dict(net.state_dict(), seed=1234, calib=calib))
net = pytorchcv_amd.set_compute_dtype(net.to(dev), "bf16")
x = synth_input(8, seed=0).to(dev).repeat((batch + 7) // 8, 1, 1, 1)[:batch].contiguous()
recs = []
orig = engine.ConvRunner._launch
def timed(self, xx, d, res, *more):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); y = orig(self, xx, d, res, *more); e.record()
    es = xx.t.element_size()
    cin_g = d.Cin // d.groups
    fl = 2.0 * y.N * y.H * y.W * d.Cout * cin_g * d.kh * d.kw
    by = (xx.N * xx.H * xx.W * d.Cin // (d.stride_h * d.stride_w if d.kh == 1 else 1) + y.N * y.H * y.W * d.Cout * (2 if res is not None else 1)) * es
    recs.append((s, e, fl, by, "%dx%d %d->%d k%d s%d g%d%s" % (xx.H, xx.W, d.Cin, d.Cout, d.kh, d.stride_h, d.groups, "+r" if res is not None else "")))
    return y
with torch.no_grad():
    for _ in range(3): net(x)
    engine.ConvRunner._launch = timed
    for _ in range(5): net(x)
    engine.ConvRunner._launch = orig
torch.cuda.synchronize()
n = len(recs) // 5
agg = {}
order = []
for i, (s, e, fl, by, tag) in enumerate(recs):
    k = (i % n, tag)
    if k not in agg: agg[k] = []; order.append(k)
    agg[k].append(s.elapsed_time(e) * 1e3)
tot = 0
for k in order:
    us = sorted(agg[k])[len(agg[k]) // 2]
    fl, by = [(r[2], r[3]) for j, r in enumerate(recs) if j % n == k[0]][0]
    tot += us
    print("%-28s %8.1f us  %6.2f TB/s  %7.1f TF" % (k[1], us, by / us / 1e6, fl / us / 1e6))
print("conv total %.1f us" % tot)
