"""
    Drop-in for the reference's `pytorchcv/model_provider.py` (get_model at :1364-1382): the registry and `get_model` of the
    MI355X-native package under the import path callers already use.
"""

from pytorchcv_amd.model_provider import *          # noqa: F401,F403
from pytorchcv_amd.model_provider import get_model, _models  # noqa: F401

__all__ = ['get_model']
