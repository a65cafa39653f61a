"""
    Declarative assembly of the classification nets: what differs between families is a stage table (widths per stage), a unit
    constructor and a handful of options - the trunk / head / variant-factory code is written once here. The attribute names
    the builder emits (`features.init_block`, `features.stage{i}.unit{j}`, `features.final_block`, `features.final_pool`,
    `output`) are the reference's state_dict contract, not a choice.
"""

__all__ = ['add_stages', 'ClassifierNet', 'register_variants', 'scale_widths', 'stage_table']

import sys
import torch.nn as nn
from ._tail import AvgPool2dNHWC, LinearHead, run_net, maybe_load_pretrained, init_conv_params, DEFAULT_ROOT


def stage_table(widths, depths):
    """[[w] * n ...]: `depths[i]` units of width `widths[i]` in stage i."""
    return [[w] * n for w, n in zip(widths, depths)]


def scale_widths(table, scale):
    """Every width of a stage table multiplied by `scale` and truncated (the zoo's width multipliers)."""
    return table if scale == 1.0 else [[int(w * scale) for w in stage] for stage in table]


def add_stages(features, in_channels, table, make_unit, container=nn.Sequential, downsample_first=False):
    """Append `stage{i+1}` containers of `unit{j+1}` modules to `features`. `make_unit(cin, cout, stride, i, j)` builds one unit;
    the first unit of every stage but the first (or of every stage with `downsample_first`) has stride 2. Returns the width
    leaving the last stage."""
    for i, widths in enumerate(table):
        stage = container()
        for j, out_channels in enumerate(widths):
            stride = 2 if j == 0 and (i != 0 or downsample_first) else 1
            stage.add_module("unit{}".format(j + 1), make_unit(in_channels, out_channels, stride, i, j))
            in_channels = out_channels
        features.add_module("stage{}".format(i + 1), stage)
    return in_channels


class ClassifierNet(nn.Module):
    """`features` (filled by the subclass through `self.trunk(...)`) + global 7x7 average pool + `output` classifier; forward =
    NCHW fp32 in, hot path, fp32 logits out (`_tail.run_net`). Subclasses call `finish(width, head=...)` last."""
    def __init__(self, in_size, num_classes):
        super(ClassifierNet, self).__init__()
        self.in_size = in_size
        self.num_classes = num_classes
        self.features = nn.Sequential()

    def finish(self, width, head=None, init=init_conv_params):
        self.features.add_module("final_pool", AvgPool2dNHWC(kernel_size=7, stride=1, fp32_out=True))
        self.output = head if head is not None else LinearHead(in_features=width, out_features=self.num_classes)
        if init is not None:
            init(self)

    def _head(self, a):
        return self.output(a)

    def forward(self, x):
        return run_net(self, x, self._head)


def register_variants(module_name, getter, variants, key="model_name"):
    """Create the zoo's named factory functions in `module_name`: for every `name: kwargs` of `variants` a function
    `name(**kwargs)` = `getter(**fixed, model_name=name, **kwargs)`, appended to the module's `__all__` (where
    `model_provider` collects the registry from)."""
    mod = sys.modules[module_name]

    def make(name, fixed):
        def factory(**kwargs):
            merged = dict(fixed)
            merged.update(kwargs)
            merged[key] = name
            return getter(**merged)
        factory.__name__ = factory.__qualname__ = name
        factory.__doc__ = "{}: {}".format(name, ", ".join("{}={!r}".format(k, v) for k, v in sorted(fixed.items())))
        return factory
    for name, fixed in variants.items():
        setattr(mod, name, make(name, fixed))
        if name not in mod.__all__:
            mod.__all__.append(name)


def pretrained_or_not(net, model_name, pretrained, root):
    return maybe_load_pretrained(net, model_name, pretrained, root)
