"""Name -> class registries with the call surface the reference's configs rely on
(``mmcv.utils.Registry`` / ``build_from_cfg``, used at reference mmseg/models/builder.py:3-31).

``@REG.register_module()`` registers under the class name; ``build_from_cfg(cfg, REG,
default_args)`` pops ``type`` and instantiates with the remaining keys, filling in
``default_args`` only where the config has no such key.  Unknown type -> KeyError.
"""
from __future__ import annotations

import inspect


class Registry:
    def __init__(self, name: str):
        self.name = name
        self._table = {}

    def __len__(self):
        return len(self._table)

    def __contains__(self, key):
        return key in self._table

    def __repr__(self):
        return f'Registry({self.name!r}, {sorted(self._table)})'

    @property
    def module_dict(self):
        return self._table

    def get(self, key):
        return self._table.get(key)

    def _add(self, cls, name, force):
        if not inspect.isclass(cls):
            raise TypeError(f'only classes can be registered, got {type(cls)}')
        key = name or cls.__name__
        if key in self._table and not force:
            raise KeyError(f'{key} is already registered in {self.name}')
        self._table[key] = cls
        return cls

    def register_module(self, name=None, force=False, module=None):
        if module is not None:
            return self._add(module, name, force)
        return lambda cls: self._add(cls, name, force)


def build_from_cfg(cfg, registry: Registry, default_args=None):
    if not isinstance(cfg, dict):
        raise TypeError(f'cfg must be a dict, got {type(cfg)}')
    if 'type' not in cfg and not (default_args and 'type' in default_args):
        raise KeyError(f'`cfg` or `default_args` must contain the key "type", got {cfg}')
    kwargs = dict(cfg)
    for k, v in (default_args or {}).items():
        kwargs.setdefault(k, v)
    kind = kwargs.pop('type')
    if isinstance(kind, str):
        cls = registry.get(kind)
        if cls is None:
            raise KeyError(f'{kind} is not in the {registry.name} registry')
    elif inspect.isclass(kind):
        cls = kind
    else:
        raise TypeError(f'type must be a str or a class, got {type(kind)}')
    return cls(**kwargs)
