"""Model discovery + `build_model`, with the reference's semantics (basicsr/models/__init__.py): `build_model(opt)`
instantiates `MODEL_REGISTRY.get(opt['model_type'])(opt)`.  Only the inference surface of `FeMaSRModel` exists here
(SURVEY 8f rank 2); training raises."""
from copy import deepcopy

from ..registry import Registry

MODEL_REGISTRY = Registry('model')

from . import femasr_model  # noqa: E402,F401  (registers FeMaSRModel)

__all__ = ['build_model', 'MODEL_REGISTRY']


def build_model(opt):
    opt = deepcopy(opt)
    return MODEL_REGISTRY.get(opt['model_type'])(opt)
