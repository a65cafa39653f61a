"""
    hipGraph capture of a whole forward. Eager execution costs one Python -> ctypes -> hipLaunchKernel trip per layer
    (~20 us each: a ResNet-50 forward is host-bound below batch ~64); a captured graph replays the same ~60 kernel
    launches from one `hipGraphLaunch`. Shapes, dtype and weights are frozen at capture time - re-capture after
    `load_state_dict`, `set_compute_dtype` or for another input shape.
"""

__all__ = ['GraphedNet', 'PipelinedNet', 'capture', 'capture_best', 'auto_lanes']

import torch


class GraphedNet(object):
    """
    g = GraphedNet(net, example)      # example: fp32 NCHW tensor on the MI355X, defines the captured shape
    y = g(x)                          # x is copied into the static input; y is the graph's static output buffer
                                      # (valid until the next call; pass clone=True for an owned copy)
    """
    def __init__(self, net: torch.nn.Module, example: torch.Tensor, warmup: int = 2, own_input: bool = False,
                 lanes: int | None = None):
        if example.device.type != "cuda":
            raise RuntimeError("graph capture needs the example input on the MI355X")
        self.net = net
        # lanes > 1: the batch is cut into `lanes` slices whose forwards are independent branches of the graph. The persistent
        # convolution kernels of one slice leave block slots idle in the last partial round of their tile schedule; the other
        # branch's kernels take those slots, so the tails overlap instead of adding up.
        # Measured (bench.py --lanes, one MI355X, bf16): +4 .. +12 % images/s at batch 128-512 on every benchmarked net with two
        # lanes, three no better, neutral below batch 64 -> `lanes=None` picks 2 from batch 64 up.
        if lanes is None:
            lanes = auto_lanes(example.shape[0])
        self.lanes = max(1, min(int(lanes), example.shape[0]))
        self._side = [torch.cuda.Stream(device=example.device) for _ in range(self.lanes - 1)]
        # own_input: `example` itself becomes the static input buffer (the caller refills it in place; no copy per call)
        self.static_in = example if (own_input and example.is_contiguous()) else example.detach().clone().contiguous()
        dev = self.static_in.device
        with torch.no_grad():
            # side stream warm-up (weight packing, allocator warm-up) as torch's capture rules require
            s = torch.cuda.Stream(device=dev)
            s.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(s):
                for _ in range(max(1, warmup)):
                    self._forward()
            torch.cuda.current_stream(dev).wait_stream(s)
            torch.cuda.synchronize(dev)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.static_out = self._forward()
        torch.cuda.synchronize(dev)

    def _forward(self):
        if self.lanes == 1:
            return self.net(self.static_in)
        cur = torch.cuda.current_stream(self.static_in.device)
        parts = self.static_in.chunk(self.lanes)
        outs = [None] * len(parts)
        for s in self._side[:len(parts) - 1]:
            s.wait_stream(cur)                               # fork
        outs[0] = self.net(parts[0])
        for i, s in enumerate(self._side[:len(parts) - 1]):
            with torch.cuda.stream(s):
                outs[i + 1] = self.net(parts[i + 1])
        for s in self._side[:len(parts) - 1]:
            cur.wait_stream(s)                               # join
        return torch.cat(outs)

    def __call__(self, x: torch.Tensor | None, clone: bool = False) -> torch.Tensor:
        """`x = None`: replay on whatever the static input holds (the caller filled it in place)."""
        if x is not None:
            if x.shape != self.static_in.shape:
                raise RuntimeError("captured for input shape {}, got {}".format(tuple(self.static_in.shape), tuple(x.shape)))
            if x.data_ptr() != self.static_in.data_ptr():
                self.static_in.copy_(x, non_blocking=True)
        self.graph.replay()
        return self.static_out.clone() if clone else self.static_out


class PipelinedNet(object):
    """
    Consecutive batches in flight: `depth` captured forwards, each with its own static input / output buffers, replayed IN TURN on
    `depth` streams. One graph on one stream starts step n+1 when step n has drained, and the tail of a forward - the 7x7 stages: a
    dozen launches of one tile round each - runs on a half-empty chip; with two steps in flight the head of the next batch fills it.
    Batch lanes inside one graph (`GraphedNet(lanes=2)`) all start together and meet their tails together; steps in flight are out of
    phase by construction. Measured (one MI355X, bench.py): MobileNetV2 batch 512 278.9k -> 299.5k img/s, ResNet-50 batch 256 83.2k ->
    85.6k, ResNeXt-101 36.9k against 37.2k with one graph of two lanes - `capture_best` times both and keeps the faster one.
        p = PipelinedNet(net, example, depth=2)
        y = p(x)            # enqueue one step on the next slot's stream; y = that slot's static output: valid once the slot's
                            # stream has caught up (p.synchronize()) and until the slot runs again, `depth` calls later
    Same kernels, same arithmetic: every slot's result is bit-identical to the eager forward of its input.
    """
    def __init__(self, net: torch.nn.Module, example: torch.Tensor, depth: int = 2, lanes: int = 1, own_input: bool = False):
        if example.device.type != "cuda":
            raise RuntimeError("graph capture needs the example input on the MI355X")
        self.depth, self.lanes = max(1, int(depth)), max(1, int(lanes))
        dev = example.device
        self._dev = dev
        self._streams = [torch.cuda.Stream(device=dev) for _ in range(self.depth)]
        self._slots = []
        cur = torch.cuda.current_stream(dev)
        for i, st in enumerate(self._streams):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                ex = example if (i == 0 and own_input) else example.detach().clone()
                self._slots.append(GraphedNet(net, ex, own_input=True, lanes=self.lanes))
        torch.cuda.synchronize(dev)
        self._next = 0

    @property
    def static_inputs(self):
        return [g.static_in for g in self._slots]

    @property
    def next_slot(self):
        return self._next

    def __call__(self, x: torch.Tensor | None = None, then=None):
        """Enqueue one step (x = None: the slot's static input as the caller left it). `then(y)`, if given, is issued on the slot's
        stream right behind the replay (a collective on the logits, a copy-out) and its result is returned instead of y."""
        k = self._next
        self._next = (k + 1) % self.depth
        g, st = self._slots[k], self._streams[k]
        st.wait_stream(torch.cuda.current_stream(self._dev))         # x (if any) was produced on the caller's stream
        with torch.cuda.stream(st):
            y = g(x)
            if then is not None:
                y = then(y)
        return y

    def synchronize(self):
        for st in self._streams:
            st.synchronize()


def auto_lanes(batch: int) -> int:
    return 2 if batch >= 64 else 1


def capture(net: torch.nn.Module, example: torch.Tensor, own_input: bool = False, lanes: int | None = None) -> GraphedNet:
    return GraphedNet(net, example, own_input=own_input, lanes=lanes)


def capture_best(net: torch.nn.Module, example: torch.Tensor, own_input: bool = False, steps: int = 24):
    """Throughput launcher for `net` at this batch: one graph of two batch lanes, or two full-batch graphs in flight - whichever
    replays `steps` steps faster here (which one wins depends on how much of the forward is single-round launches). Below batch 64
    there is nothing to choose: one graph, one lane. Returns a GraphedNet or a PipelinedNet (both: `obj(x_or_None)`)."""
    import time
    if example.shape[0] < 64:
        return GraphedNet(net, example, own_input=own_input, lanes=1)
    dev = example.device
    cands = [GraphedNet(net, example, own_input=own_input, lanes=2), PipelinedNet(net, example, depth=2, lanes=1, own_input=False)]
    best, best_t = None, None
    times = [[] for _ in cands]
    for rnd in range(3):                                             # interleaved rounds: clocks and caches drift over a measurement
        for i, c in enumerate(cands):
            for _ in range(3):
                c(None)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(steps // 3):
                c(None)
            torch.cuda.synchronize(dev)
            times[i].append(time.perf_counter() - t0)
    names = ["one graph, two batch lanes", "two graphs in flight, one lane each"]
    report = {}
    for i, c in enumerate(cands):
        t = min(times[i])
        report[names[i]] = round(example.shape[0] * (steps // 3) / t, 1)          # images/s of the short measurement
        if best_t is None or t < best_t * (0.995 if i > 0 else 1.0):  # the second launcher doubles the activation memory: it has to WIN
            best, best_t = c, t
    del cands
    best.tuning = report                                             # what the choice was made on (bench.py prints it)
    return best
