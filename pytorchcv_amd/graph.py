"""
    hipGraph capture of a whole forward. Eager execution costs one Python -> ctypes -> hipLaunchKernel trip per layer
    (~20 us each: a ResNet-50 forward is host-bound below batch ~64); a captured graph replays the same ~60 kernel
    launches from one `hipGraphLaunch`. Shapes, dtype and weights are frozen at capture time - re-capture after
    `load_state_dict`, `set_compute_dtype` or for another input shape.
"""

__all__ = ['GraphedNet', 'capture', 'auto_lanes']

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

    def __call__(self, x: torch.Tensor, clone: bool = False) -> torch.Tensor:
        if x.shape != self.static_in.shape:
            raise RuntimeError("captured for input shape {}, got {}".format(tuple(self.static_in.shape), tuple(x.shape)))
        if x.data_ptr() != self.static_in.data_ptr():
            self.static_in.copy_(x, non_blocking=True)
        self.graph.replay()
        return self.static_out.clone() if clone else self.static_out


def auto_lanes(batch: int) -> int:
    return 2 if batch >= 64 else 1


def capture(net: torch.nn.Module, example: torch.Tensor, own_input: bool = False, lanes: int | None = None) -> GraphedNet:
    return GraphedNet(net, example, own_input=own_input, lanes=lanes)
