"""TEST INFRASTRUCTURE ONLY (oracle/): import shim for the upstream reference.

Used ONLY by oracle/gen_golden.py, in the build container, to run the real reference
code on CPU and dump golden vectors into tests/golden/.  /root/reference does not exist
on the GPU box and nothing under tests/ -m gpu, smoke() or bench.py imports this file.

The shim: stub modules for imports the reference makes but never calls on
the Learner.fit() hot path (seaborn, spacy, cv2, skimage, GPUtil, IPython, torchvision,
pycocotools), `.cuda()` turned into identity (the reference hard-codes `.cuda()`,
General/Learner.py:107, General/Core.py:70,140-144), bytecode writing disabled.
"""
import sys, types, importlib

REF = '/root/reference'


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install():
    sys.dont_write_bytecode = True
    import torch, torch.nn as nn, tqdm
    _stub('seaborn')
    sp = _stub('spacy'); _stub('spacy.symbols', ORTH=0); sp.symbols = sys.modules['spacy.symbols']
    _stub('cv2')
    sk = _stub('skimage'); sk.io = _stub('skimage.io'); sk.transform = _stub('skimage.transform')
    _stub('GPUtil')
    ip = _stub('IPython', get_ipython=lambda: None, version_info=(0, 0)); ip.display = _stub('IPython.display', clear_output=lambda *a, **k: None)

    class _ResNet(nn.Module):
        pass
    tv = _stub('torchvision'); tv.models = _stub('torchvision.models', ResNet=_ResNet)
    tv.transforms = _stub('torchvision.transforms')
    pc = _stub('pycocotools'); pc._mask = _stub('pycocotools._mask', iou=None, merge=None, frPyObjects=None)
    tqdm.tqdm_notebook = lambda it=None, *a, **k: it
    torch.Tensor.cuda = lambda self, *a, **k: self
    nn.Module.cuda = lambda self, *a, **k: self
    torch.cuda.empty_cache = lambda: None
    if REF not in sys.path:
        sys.path.insert(0, REF)


def load():
    install()
    mods = {}
    for name in ['General.Core', 'General.Layers', 'General.Optimizer', 'General.LossesMetrics',
                 'General.Learner', 'Applications.CollabFiltering', 'Applications.StructuredData',
                 'Applications.VisionModels.retinanet', 'Applications.Vision', 'Applications.Text']:
        mods[name] = importlib.import_module(name)
    return mods


if __name__ == '__main__':
    m = load()
    print('reference imported:', sorted(m))
