"""Checkpoint I/O: load reference-format checkpoints ({'state_dict': ...} or a bare state
dict, optional 'module.' prefix) key-for-key; save student-only training state with the
distillation step counter (the reference loses it on resume)."""
from __future__ import annotations

import os

import torch


def _state_dict_of(obj):
    if isinstance(obj, dict):
        for k in ('state_dict', 'model'):
            if k in obj and isinstance(obj[k], dict):
                return obj[k]
    return obj


def load_checkpoint(module, path, strict=False, prefix=None, map_location='cpu'):
    if not os.path.isfile(path):
        raise FileNotFoundError(f'checkpoint {path} does not exist')
    sd = _state_dict_of(torch.load(path, map_location=map_location, weights_only=False))
    if all(k.startswith('module.') for k in sd):
        sd = {k[7:]: v for k, v in sd.items()}
    if prefix:
        sd = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    result = module.load_state_dict(sd, strict=strict)
    return result


def save_checkpoint(path, model, optimizer=None, meta=None):
    state = {'state_dict': model.state_dict(), 'meta': dict(meta or {})}
    if optimizer is not None:
        state['optimizer'] = optimizer.state_dict()
    tmp = path + '.tmp'
    torch.save(state, tmp)
    os.replace(tmp, path)
