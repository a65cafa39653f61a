"""Dev experiment: consecutive batches in flight. One hipGraph per step replays on ONE stream, so step n+1 starts when step n has
drained - the latency-bound 7x7 tail of a forward runs on a half-empty chip. Here two (or three) captured graphs with their own static
buffers replay on different streams in turn: the head of step n+1 overlaps the tail of step n (phase diversity, which batch lanes
inside one graph do not have: they all start together).

    python tests/tools/exp_pipeline.py <model> <batch> [dtype]
"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import pytorchcv_amd
from pytorchcv_amd.graph import capture
from pytorchcv_amd.model_provider import get_model
from pytorchcv_amd.synth import synth_state_dict, synth_input

name, batch = sys.argv[1], int(sys.argv[2])
dtype = sys.argv[3] if len(sys.argv) > 3 else "auto"
dev = torch.device("cuda", 0)
net = get_model(name).eval()
cal = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "golden", "calib_%s.json" % name)
calib = {k: tuple(v) for k, v in json.load(open(cal)).items()} if os.path.exists(cal) else None
net.load_state_dict(synth_state_dict(net.state_dict(), seed=1234, calib=calib))
net = pytorchcv_amd.set_compute_dtype(net.to(dev), dtype)
x = synth_input(8, seed=0).to(dev).repeat((batch + 7) // 8, 1, 1, 1)[:batch].contiguous()


def run(depth, lanes, steps=40, warmup=6, offset_ms=0.0):
    streams = [torch.cuda.Stream(device=dev) for _ in range(depth)]
    graphs = []
    for s in streams:
        with torch.cuda.stream(s):
            graphs.append(capture(net, x.clone(), own_input=True, lanes=lanes))
    torch.cuda.synchronize()
    outs = [None] * depth
    def loop(n):
        for i in range(n):
            k = i % depth
            with torch.cuda.stream(streams[k]):
                outs[k] = graphs[k](graphs[k].static_in)
    loop(warmup)
    torch.cuda.synchronize()
    if offset_ms > 0 and depth > 1:                                  # hold the second stream back once: a deliberate phase offset
        with torch.cuda.stream(streams[1]):
            torch.cuda._sleep(int(offset_ms * 1e-3 * 100e6))         # _sleep counts the 100 MHz wall clock of the device
    t0 = time.perf_counter()
    loop(steps)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    same = all(torch.equal(outs[0], o) for o in outs[1:])
    return batch / dt, dt * 1e3, same


for depth, lanes in ((1, 2), (2, 1), (2, 2), (3, 1), (3, 2)):
    v, ms, same = run(depth, lanes)
    print("%s bs%d %s: %d graph(s) in flight x %d lane(s): %9.1f img/s  %.3f ms/step  outputs equal: %s" % (name, batch, dtype, depth, lanes, v, ms, same), flush=True)
v0, ms0, _ = run(2, 1)
for frac in (0.25, 0.5, 0.75):
    v, ms, same = run(2, 1, offset_ms=frac * ms0)
    print("%s bs%d %s: 2 in flight x 1 lane, second stream held back %.2f of a step once: %9.1f img/s  %.3f ms/step (no hold: %.1f)" % (name, batch, dtype, frac, v, ms, v0), flush=True)
