"""
    pytorchcv_amd - MI355X (gfx950) native conv-net inference path behind the pytorchcv `get_model` API.
"""

__version__ = "0.1.0"

from .engine import set_compute_dtype, NHWC  # noqa: F401
