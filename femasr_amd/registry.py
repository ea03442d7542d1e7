"""Name -> class registry with BasicSR semantics.

Mirrors the behaviour (not the code) of the reference registry
(basicsr/utils/registry.py:4-82): objects are keyed by ``__name__``,
registering a duplicate name is an error, ``get`` raises ``KeyError`` for an
unknown name, and ``register`` works both as a decorator and as a plain call.
"""


class Registry:
    def __init__(self, name):
        self._name = name
        self._obj_map = {}

    def _add(self, obj):
        key = obj.__name__
        if key in self._obj_map:
            raise AssertionError(
                f"An object named '{key}' was already registered in '{self._name}' registry!")
        self._obj_map[key] = obj
        return obj

    def register(self, obj=None):
        # decorator form: @REG.register()    call form: REG.register(cls)
        if obj is None:
            return self._add
        self._add(obj)

    def get(self, name):
        try:
            return self._obj_map[name]
        except KeyError:
            raise KeyError(f"No object named '{name}' found in '{self._name}' registry!") from None

    def __contains__(self, name):
        return name in self._obj_map

    def __iter__(self):
        return iter(self._obj_map.items())

    def keys(self):
        return self._obj_map.keys()


ARCH_REGISTRY = Registry('arch')
