"""
    Normalisation generators (reference pytorchcv/models/common/norm.py:34-50,95-115). Only eval-mode BatchNorm2d is on the
    hot path: it holds the parameters under the reference's state_dict names and is folded into the convolution
    epilogue as fp32 scale/shift (pcv_bn_fold).
"""

__all__ = ['lambda_batchnorm2d', 'create_normalization_layer']

from inspect import isfunction
import torch.nn as nn


def lambda_batchnorm2d(eps: float = 1e-5):
    return lambda num_features: nn.BatchNorm2d(num_features=num_features, eps=eps)


def create_normalization_layer(normalization, **kwargs):
    assert (normalization is not None)
    if isfunction(normalization):
        return normalization(**kwargs)
    assert isinstance(normalization, nn.Module)
    return normalization
