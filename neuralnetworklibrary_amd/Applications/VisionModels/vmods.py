"""Model registry of the drop-in API (mirror of Applications/VisionModels/vmods.py).  The reference also vendors the
Cadene senet / nasnet / inception / resnext zoo (out of scope: not on any BASELINE config);
their class names exist here as placeholders so that `default_cut` / `default_split` isinstance checks keep working."""
from . import retinanet, resnet  # noqa: F401
from .resnet import ResNet, resnet18, resnet34, resnet50, resnet101, resnet152  # noqa: F401


class _NotVendored:
    "placeholder type for a vendored-zoo architecture that this build does not ship"


ResNeXt101_32x4d = type('ResNeXt101_32x4d', (_NotVendored,), {})
ResNeXt101_64x4d = type('ResNeXt101_64x4d', (_NotVendored,), {})
SENet = type('SENet', (_NotVendored,), {})
InceptionV4 = type('InceptionV4', (_NotVendored,), {})
