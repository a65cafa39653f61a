"""
    Activation generators with the reference's names and call conventions (pytorchcv/models/common/activ.py:50-222).
    The modules are parameter-free markers: on the hot path an activation is a code in the convolution epilogue
    (engine.act_code), it never runs as a separate pass.
"""

__all__ = ['Swish', 'HSigmoid', 'HSwish', 'lambda_relu', 'lambda_relu6', 'lambda_sigmoid', 'lambda_swish', 'lambda_hsigmoid',
           'lambda_hswish', 'create_activation_layer']

from inspect import isfunction
import torch.nn as nn


class _EpilogueOnly(nn.Module):
    def forward(self, x):
        raise RuntimeError("{} is applied inside the fused convolution epilogue on MI355X and cannot be called on its own"
                           .format(type(self).__name__))


class Swish(_EpilogueOnly):
    """x * sigmoid(x) (reference activ.py:16-21)."""


class HSigmoid(_EpilogueOnly):
    """relu6(x + 3) / 6 (reference activ.py:24-30)."""


class HSwish(_EpilogueOnly):
    """x * relu6(x + 3) / 6 (reference activ.py:33-47)."""
    def __init__(self, inplace: bool = False):
        super(HSwish, self).__init__()
        self.inplace = inplace


def lambda_relu(inplace: bool = True):
    return lambda: nn.ReLU(inplace=inplace)


def lambda_relu6(inplace: bool = True):
    return lambda: nn.ReLU6(inplace=inplace)


def lambda_sigmoid():
    return lambda: nn.Sigmoid()


def lambda_swish():
    return lambda: Swish()


def lambda_hsigmoid():
    return lambda: HSigmoid()


def lambda_hswish(inplace: bool = True):
    return lambda: HSwish(inplace=inplace)


_BY_NAME = {"relu": lambda: nn.ReLU(inplace=True), "relu6": lambda: nn.ReLU6(inplace=True), "swish": Swish,
            "hswish": lambda: HSwish(inplace=True), "sigmoid": nn.Sigmoid, "hsigmoid": HSigmoid}


def create_activation_layer(activation):
    """Lambda generator, name or module -> module (same three input kinds as reference activ.py:188-222)."""
    assert (activation is not None)
    if isfunction(activation):
        return activation()
    if isinstance(activation, str):
        if activation not in _BY_NAME:
            raise NotImplementedError()
        return _BY_NAME[activation]()
    assert isinstance(activation, nn.Module)
    return activation
