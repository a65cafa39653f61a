"""
    `get_model(name, **kwargs)` with the reference's contract (pytorchcv/model_provider.py:1364-1382): case-insensitive
    name, `ValueError("Unsupported model: ...")` for unknown names, kwargs (`pretrained`, `root`, `in_channels`, `in_size`,
    `num_classes`) forwarded to the factory. The registry holds the families whose whole forward runs on the MI355X hot
    path (ResNet, SE-ResNet, ResNeXt, SE-ResNeXt, MobileNet, MobileNetV2, MobileNetV3, EfficientNet, PreResNet, SE-PreResNet, DenseNet, ShuffleNetV2).
"""

__all__ = ['get_model']

from .models import resnet as _resnet
from .models import mobilenetv2 as _mobilenetv2
from .models import resnext as _resnext
from .models import seresnet as _seresnet
from .models import seresnext as _seresnext
from .models import mobilenet as _mobilenet
from .models import mobilenetv3 as _mobilenetv3
from .models import efficientnet as _efficientnet
from .models import preresnet as _preresnet
from .models import sepreresnet as _sepreresnet
from .models import densenet as _densenet
from .models import shufflenetv2 as _shufflenetv2
from .models import vgg as _vgg

_models = {}
for _mod in (_resnet, _mobilenetv2, _resnext, _seresnet, _seresnext, _mobilenet, _mobilenetv3, _efficientnet, _preresnet, _sepreresnet, _densenet, _shufflenetv2, _vgg):
    for _name in _mod.__all__:
        _fn = getattr(_mod, _name)
        if _name.islower() and not _name.startswith(("get_", "calc_")) and callable(_fn):
            _models[_name] = _fn


def get_model(name, **kwargs):
    name = name.lower()
    if name not in _models:
        raise ValueError("Unsupported model: {}".format(name))
    net = _models[name](**kwargs)
    return net
