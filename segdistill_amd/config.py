"""Python-file configs with the merge semantics the reference's experiment files use
(``mmcv.Config.fromfile``, called at reference tools/train.py:67-69):

* the file is executed; its public module-level names become keys;
* ``_base_`` (a path or a list of paths, relative to the file) is loaded first and
  deep-merged, the child winning; a child dict carrying ``_delete_=True`` replaces the
  base value instead of merging into it;
* ``merge_from_dict({'a.b': v})`` applies ``--options`` style overrides;
* dict values support attribute access (``cfg.model.type``).
"""
from __future__ import annotations

import copy
import os
import types

DELETE_KEY = '_delete_'
BASE_KEY = '_base_'


class ConfigDict(dict):
    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError as e:
            raise AttributeError(name) from e

    def __setattr__(self, name, value):
        self[name] = value

    def __deepcopy__(self, memo):
        return ConfigDict({k: copy.deepcopy(v, memo) for k, v in self.items()})


def _wrap(v):
    if isinstance(v, dict):
        return ConfigDict({k: _wrap(x) for k, x in v.items()})
    if isinstance(v, list):
        return [_wrap(x) for x in v]
    if isinstance(v, tuple):
        return tuple(_wrap(x) for x in v)
    return v


def _merge(child, base):
    """Merge ``child`` INTO a copy of ``base`` (child wins)."""
    out = dict(base)
    for k, v in child.items():
        if isinstance(v, dict) and k in out and isinstance(out[k], dict) and not v.get(DELETE_KEY, False):
            out[k] = _merge(v, out[k])
        elif isinstance(v, dict):
            v = dict(v)
            v.pop(DELETE_KEY, None)
            out[k] = v
        else:
            out[k] = v
    return out


def _exec_file(path):
    with open(path, 'r', encoding='utf-8') as f:
        src = f.read()
    scope = {'__file__': path, '__name__': '__segdistill_config__'}
    exec(compile(src, path, 'exec'), scope)  # configs are trusted user code, as with mmcv
    out = {}
    for k, v in scope.items():
        if k.startswith('__') or isinstance(v, (types.ModuleType, types.FunctionType)) or isinstance(v, type):
            continue
        out[k] = v
    return out


def _load(path):
    path = os.path.abspath(os.path.expanduser(path))
    if not os.path.isfile(path):
        raise FileNotFoundError(path)
    own = _exec_file(path)
    bases = own.pop(BASE_KEY, None)
    if bases is None:
        return own
    if isinstance(bases, str):
        bases = [bases]
    merged = {}
    for b in bases:
        sub = _load(os.path.join(os.path.dirname(path), b))
        dup = merged.keys() & sub.keys()
        if dup:
            raise KeyError(f'Duplicate key is not allowed among bases: {sorted(dup)}')
        merged.update(sub)
    return _merge(own, merged)


class Config:
    def __init__(self, cfg_dict=None, filename=None):
        object.__setattr__(self, '_cfg', _wrap(cfg_dict or {}))
        object.__setattr__(self, 'filename', filename)

    @staticmethod
    def fromfile(filename):
        return Config(_load(filename), filename=filename)

    def __getattr__(self, name):
        return getattr(self._cfg, name)

    def __getitem__(self, name):
        return self._cfg[name]

    def __setattr__(self, name, value):
        self._cfg[name] = _wrap(value)

    def __setitem__(self, name, value):
        self._cfg[name] = _wrap(value)

    def __contains__(self, name):
        return name in self._cfg

    def get(self, key, default=None):
        return self._cfg.get(key, default)

    def keys(self):
        return self._cfg.keys()

    def to_dict(self):
        return copy.deepcopy(dict(self._cfg))

    def merge_from_dict(self, options):
        nested = {}
        for full, v in options.items():
            d = nested
            parts = full.split('.')
            for p in parts[:-1]:
                d = d.setdefault(p, {})
            d[parts[-1]] = v
        object.__setattr__(self, '_cfg', _wrap(_merge(nested, self._cfg)))
