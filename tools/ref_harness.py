"""Import harness for the READ-ONLY reference at /root/reference.

Test-infrastructure glue only: it runs exclusively in the development
container (the reference does not travel to the GPU box) and is used by
`tests/golden/make_golden.py` to produce the committed golden vectors.
Nothing under `femasr_amd/` imports this file.

Why stubs: `basicsr/__init__.py:3-9` star-imports training code needing cv2 /
pyiqa / lmdb, `femasr_arch.py:11` imports vgg_arch (torchvision) and
`network_swinir.py:11` imports timm — none installed here.  The stubs are
construction-time only (DropPath is Identity at rate 0, network_swinir.py:204);
no arithmetic on the hot path comes from them.
"""
import importlib
import sys
import types

import torch.nn as nn

REF_ROOT = '/root/reference'


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _DropPath(nn.Module):
    def __init__(self, p=0.):
        super().__init__()

    def forward(self, x):
        return x


def import_reference_arch():
    """Returns the reference module `basicsr.archs.femasr_arch`."""
    if 'basicsr.archs.femasr_arch' in sys.modules:
        return sys.modules['basicsr.archs.femasr_arch']
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    _stub('timm')
    _stub('timm.models')
    _stub('timm.models.layers', DropPath=_DropPath,
          to_2tuple=lambda x: tuple(x) if isinstance(x, (tuple, list)) else (x, x),
          trunc_normal_=nn.init.trunc_normal_)
    tv = _stub('torchvision', __version__='stub')
    tv.models = _stub('torchvision.models', vgg=types.SimpleNamespace())
    tv.utils = _stub('torchvision.utils', make_grid=None)
    for name in ('cv2', 'pyiqa', 'lmdb'):
        _stub(name)
    pkg = types.ModuleType('basicsr')
    pkg.__path__ = [REF_ROOT + '/basicsr']      # bypass basicsr/__init__.py
    sys.modules['basicsr'] = pkg
    return importlib.import_module('basicsr.archs.femasr_arch')


def import_reference_swin():
    import_reference_arch()
    return importlib.import_module('basicsr.archs.network_swinir')
