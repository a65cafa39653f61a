"""
    hipGraph capture of a whole forward. Eager execution costs one Python -> ctypes -> hipLaunchKernel trip per layer
    (~20 us each: a ResNet-50 forward is host-bound below batch ~64); a captured graph replays the same ~60 kernel
    launches from one `hipGraphLaunch`. Shapes, dtype and weights are frozen at capture time - re-capture after
    `load_state_dict`, `set_compute_dtype` or for another input shape.
"""

__all__ = ['GraphedNet', 'capture']

import torch


class GraphedNet(object):
    """
    g = GraphedNet(net, example)      # example: fp32 NCHW tensor on the MI355X, defines the captured shape
    y = g(x)                          # x is copied into the static input; y is the graph's static output buffer
                                      # (valid until the next call; pass clone=True for an owned copy)
    """
    def __init__(self, net: torch.nn.Module, example: torch.Tensor, warmup: int = 2, own_input: bool = False):
        if example.device.type != "cuda":
            raise RuntimeError("graph capture needs the example input on the MI355X")
        self.net = net
        # own_input: `example` itself becomes the static input buffer (the caller refills it in place; no copy per call)
        self.static_in = example if (own_input and example.is_contiguous()) else example.detach().clone().contiguous()
        dev = self.static_in.device
        with torch.no_grad():
            # side stream warm-up (weight packing, allocator warm-up) as torch's capture rules require
            s = torch.cuda.Stream(device=dev)
            s.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(s):
                for _ in range(max(1, warmup)):
                    net(self.static_in)
            torch.cuda.current_stream(dev).wait_stream(s)
            torch.cuda.synchronize(dev)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.static_out = net(self.static_in)
        torch.cuda.synchronize(dev)

    def __call__(self, x: torch.Tensor, clone: bool = False) -> torch.Tensor:
        if x.shape != self.static_in.shape:
            raise RuntimeError("captured for input shape {}, got {}".format(tuple(self.static_in.shape), tuple(x.shape)))
        if x.data_ptr() != self.static_in.data_ptr():
            self.static_in.copy_(x, non_blocking=True)
        self.graph.replay()
        return self.static_out.clone() if clone else self.static_out


def capture(net: torch.nn.Module, example: torch.Tensor, own_input: bool = False) -> GraphedNet:
    return GraphedNet(net, example, own_input=own_input)
