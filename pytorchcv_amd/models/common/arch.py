"""
    Branch / merge containers of the reference (pytorchcv/models/common/arch.py:58-131) on the NHWC hot path.

    `Concurrent` runs every child on the same input and merges the results; `SequentialConcurrent` chains the children and
    merges all intermediate results. On the GPU path a channel concatenation is not an op: the merged tensor is allocated once
    and every branch that ends in a ConvBlock writes its channel slice from the convolution epilogue (`y_cpitch`,
    `ConvRunner.run(out=...)`); a summing merge rides in the epilogue as the residual operand of each branch's last convolution.
    Branches that end in something else are merged by one copy / add pass. The branch output shapes are learnt on the first
    forward of an input shape (run unmerged once), every later forward of that shape is merge-free.
"""

__all__ = ['Concurrent', 'SequentialConcurrent', 'ParametricSequential', 'ParametricConcurrent']

import torch
import torch.nn as nn
from ... import engine
from .conv import ConvBlock


def _tail_conv(branch):
    """(modules before, last ConvBlock) when the branch is a ConvBlock or an nn.Sequential that ends in one and that block can
    write a channel slice / take a residual (dense convolution, no per-call padding), else None."""
    head, tail = [], branch
    if isinstance(branch, nn.Sequential) and not isinstance(branch, (Concurrent, SequentialConcurrent)) and len(branch) > 0:
        mods = list(branch.children())
        head, tail = mods[:-1], mods[-1]
    if isinstance(tail, ConvBlock) and tail.conv.out_channels % 8 == 0 and not (
            tail.conv.groups > 1 and tail.conv.groups == tail.conv.in_channels == tail.conv.out_channels) and not (
            tail.conv.in_channels <= 4 and max(tail.conv.stride) == 2):       # a stem-shaped convolution: its kernel writes a dense y only
        return head, tail
    return None


def _run_head(mods, a):
    for m in mods:
        a = m(a)
    return a


def _stack_nchw(outs, axis):
    return torch.stack(tuple(engine.to_nchw(o) if isinstance(o, engine.NHWC) else o for o in outs), dim=axis)


def _merge_by_copy(outs):
    """torch.cat(outs, dim=1) for NHWC handles whose channel counts are multiples of 8 (copy form)."""
    first = outs[0]
    if any((o.N, o.H, o.W) != (first.N, first.H, first.W) or o.dtype != first.dtype for o in outs):
        raise RuntimeError("Concurrent: branch outputs differ in shape: {}".format([o.size() for o in outs]))
    total = sum(o.C for o in outs)
    if any(o.C % 8 for o in outs):
        # odd widths: through the NCHW boundary (exact, slow; the zoo's concatenating blocks use multiples of 8)
        x = torch.cat(tuple(engine.to_nchw(o) for o in outs), dim=1)
        return engine.from_nchw(x, {v[1]: k for k, v in engine.DTYPES.items()}[first.dtype], stem=False)
    buf = torch.empty((first.N, first.H, first.W, total), dtype=first.dtype, device=first.device)
    off = 0
    for o in outs:
        engine.channel_concat_into(o, buf, off)
        off += o.C
    return engine.NHWC(buf, first.N, first.H, first.W, total)


class Concurrent(nn.Sequential):
    """Every child sees the same input; the outputs are concatenated along `axis` ("cat", the default), stacked along a new
    axis ("stack" / `stack=True`) or summed ("sum"). Same constructor and child registration as the reference's container."""
    def __init__(self, axis=1, stack=False, merge_type=None):
        super(Concurrent, self).__init__()
        assert (merge_type is None) or (merge_type in ["cat", "stack", "sum"])
        self.axis = axis
        self.merge_type = merge_type if merge_type is not None else ("stack" if stack else "cat")
        self._pcv_shapes = {}

    def _run(self, a):
        branches = list(self.children())
        if self.merge_type == "stack":
            return _stack_nchw([m(a) for m in branches], self.axis)
        if self.merge_type == "sum":
            acc = None
            for m in branches:
                tc = _tail_conv(m) if acc is not None else None
                if tc is not None:
                    acc = tc[1](_run_head(tc[0], a), residual=acc)            # act(BN(conv)) + running sum, one launch
                else:
                    y = m(a)
                    acc = y if acc is None else engine.add(acc, y)
            return acc
        if self.axis != 1:
            raise NotImplementedError("Concurrent: concatenation along axis {} is not on the MI355X path".format(self.axis))
        # the learnt branch shapes belong to THESE children: a child added or replaced later starts over
        key = (a.N, a.H, a.W, a.C, a.dtype, tuple(id(m) for m in branches))
        shapes = self._pcv_shapes.get(key)
        if shapes is None:                                                  # first forward of this shape: learn the branch outputs
            outs = [m(a) for m in branches]
            y = _merge_by_copy(outs)
            if all(o.C % 8 == 0 for o in outs):
                self._pcv_shapes[key] = [(o.H, o.W, o.C) for o in outs]
            return y
        assert len(shapes) == len(branches)
        H, W = shapes[0][0], shapes[0][1]
        total = sum(c for _, _, c in shapes)
        buf = torch.empty((a.N, H, W, total), dtype=a.dtype, device=a.device)
        off = 0
        for m, (_, _, c) in zip(branches, shapes):
            tc = _tail_conv(m)
            if tc is not None and tc[1].conv.out_channels == c:
                tc[1](_run_head(tc[0], a), out=(buf, off))                   # the convolution writes its slice: no copy
            else:
                engine.channel_concat_into(m(a), buf, off)
            off += c
        return engine.NHWC(buf, a.N, H, W, total)

    def forward(self, x):
        return engine.boundary(self, x, self._run)


class SequentialConcurrent(nn.Sequential):
    """The children run one after the other; the input (when `cat_input`) and every intermediate output are concatenated along
    `axis` (or stacked). Each output is also the next child's input, so the merge is the copy form."""
    def __init__(self, axis=1, stack=False, cat_input=True):
        super(SequentialConcurrent, self).__init__()
        self.axis = axis
        self.stack = stack
        self.cat_input = cat_input

    def _run(self, a):
        outs = [a] if self.cat_input else []
        for m in self.children():
            a = m(a)
            outs.append(a)
        if self.stack:
            return _stack_nchw(outs, self.axis)
        if self.axis != 1:
            raise NotImplementedError("SequentialConcurrent: concatenation along axis {} is not on the MI355X path".format(self.axis))
        return _merge_by_copy(outs)

    def forward(self, x):
        return engine.boundary(self, x, self._run)


class ParametricSequential(nn.Sequential):
    """nn.Sequential whose children all receive the same keyword arguments."""
    def forward(self, x, **kwargs):
        for m in self.children():
            x = m(x, **kwargs)
        return x


class ParametricConcurrent(nn.Sequential):
    """`Concurrent` (channel concatenation) whose children all receive the same keyword arguments."""
    def __init__(self, axis=1):
        super(ParametricConcurrent, self).__init__()
        self.axis = axis

    def forward(self, x, **kwargs):
        def run(a):
            if self.axis != 1:
                raise NotImplementedError("ParametricConcurrent: concatenation along axis {} is not on the MI355X path".format(self.axis))
            return _merge_by_copy([m(a, **kwargs) for m in self.children()])
        return engine.boundary(self, x, run)
