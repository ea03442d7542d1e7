"""Arch discovery + `build_network`, with the reference's semantics
(basicsr/archs/__init__.py:13-25): every `*_arch.py` in this folder is imported
so its classes register themselves, and `build_network(opt)` pops `type` and
instantiates `ARCH_REGISTRY.get(type)(**opt)` — the YAML `network_g` block
(options/train_FeMaSR_LQ_stage.yml:45-55) is passed through verbatim."""
import importlib
import os
from copy import deepcopy

from ..registry import ARCH_REGISTRY

__all__ = ['build_network', 'ARCH_REGISTRY']

_folder = os.path.dirname(os.path.abspath(__file__))
_arch_modules = [importlib.import_module(f'{__name__}.{f[:-3]}')
                 for f in sorted(os.listdir(_folder)) if f.endswith('_arch.py')]


def build_network(opt):
    opt = deepcopy(opt)
    network_type = opt.pop('type')
    return ARCH_REGISTRY.get(network_type)(**opt)
