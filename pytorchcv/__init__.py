"""
    `pytorchcv` alias of the MI355X-native package `pytorchcv_amd`, so that callers of the reference keep their imports:

        from pytorchcv.model_provider import get_model          # reference pytorchcv/model_provider.py:1364-1382
        from pytorchcv.models.resnet import resnet50            # reference pytorchcv/models/resnet.py
        from pytorchcv.models.common.model_store import load_model, calc_net_weight_count

    Nothing is defined here: every `pytorchcv.X` module IS the `pytorchcv_amd.X` module object (one set of classes, one
    model registry, one weight store), resolved on import by the finder below. `pytorchcv/model_provider.py` exists as a
    file only so that the entry point named by the drop-in contract can be read where a maintainer looks for it.
"""

import os
import sys
import importlib
import importlib.abc
import importlib.util

import pytorchcv_amd as _impl

__version__ = getattr(_impl, "__version__", "0.0.73+amd")
_PREFIX, _TARGET = __name__ + ".", _impl.__name__ + "."
_HERE = os.path.dirname(os.path.abspath(__file__))


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    """`pytorchcv.a.b` -> the already-importable `pytorchcv_amd.a.b` (same module object, registered under both names)."""

    def find_spec(self, fullname, path=None, target=None):
        if not fullname.startswith(_PREFIX):
            return None
        rel = fullname[len(_PREFIX):].replace(".", os.sep)
        if os.path.exists(os.path.join(_HERE, rel + ".py")) or os.path.isdir(os.path.join(_HERE, rel)):
            return None                                          # a file of this package (model_provider.py): the stock finders load it
        real = _TARGET + fullname[len(_PREFIX):]
        try:
            if importlib.util.find_spec(real) is None:
                return None
        except (ImportError, ValueError):
            return None
        return importlib.util.spec_from_loader(fullname, self)

    def create_module(self, spec):
        return importlib.import_module(_TARGET + spec.name[len(_PREFIX):])

    def exec_module(self, module):       # the real module is already initialised
        pass


# in FRONT of the path finders: the aliased parent package's __path__ is pytorchcv_amd's, where the stock finder would load a
# second copy of a submodule under the `pytorchcv.` name (two sets of classes) before this finder is asked
if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _AliasFinder())

set_compute_dtype = _impl.set_compute_dtype
