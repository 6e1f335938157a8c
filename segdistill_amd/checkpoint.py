"""Checkpoint I/O: load reference-format checkpoints ({'state_dict': ...} / {'model': ...} or a bare state dict, optional
'module.' prefix) key-for-key, with the adaptations the reference's Swin loader applies to public checkpoints
(mmcv_custom/checkpoint.py:281-347: MoBY 'encoder.' branch, absolute-position-embedding layout, bicubic resize of the
relative-position-bias tables when the window size differs); save student-only training state with the distillation step
counter (the reference loses it on resume)."""
from __future__ import annotations

import os

import torch


def _state_dict_of(obj):
    if isinstance(obj, dict):
        for k in ('state_dict', 'model', 'student'):
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
    sd = adapt_public_state_dict(module, dict(sd))
    own = module.state_dict()
    if own and sd and not any(k in own for k in sd):
        # strict=False would "succeed" while loading nothing at all (e.g. a whole-SDModule file given without prefix='student.')
        raise KeyError(f'{path}: none of its {len(sd)} keys (e.g. {sorted(sd)[0]!r}) names a parameter of {type(module).__name__} '
                       f'(e.g. {next(iter(own))!r}); wrong checkpoint or missing `prefix`')
    result = module.load_state_dict(sd, strict=strict)
    return result


def adapt_public_state_dict(module, sd):
    """mmcv_custom/checkpoint.py:314-343, applied only where the keys exist (a no-op for MiT / ResNet checkpoints)."""
    import warnings

    import torch.nn.functional as F
    if sd and sorted(sd)[0].startswith('encoder'):                       # MoBY: keep the online branch (:314-316)
        sd = {k.replace('encoder.', ''): v for k, v in sd.items() if k.startswith('encoder.')}
    own = module.state_dict()
    ape = sd.get('absolute_pos_embed')
    if ape is not None and 'absolute_pos_embed' in own and ape.dim() == 3:   # [1, L, C] -> [1, C, H, W] (:318-326)
        n1, l1, c1 = ape.shape
        n2, c2, h, w = own['absolute_pos_embed'].shape
        if n1 != n2 or c1 != c2 or l1 != h * w:
            warnings.warn('absolute_pos_embed of the checkpoint does not fit the model: skipped')
            sd.pop('absolute_pos_embed')
        else:
            sd['absolute_pos_embed'] = ape.view(n2, h, w, c2).permute(0, 3, 1, 2)
    for key in [k for k in sd if 'relative_position_bias_table' in k]:       # window-size change (:328-343)
        if key not in own:
            continue
        src, dst = sd[key], own[key]
        (l1, h1), (l2, h2) = src.shape, dst.shape
        if h1 != h2:
            warnings.warn(f'{key}: {h1} heads in the checkpoint, {h2} in the model: skipped')
            sd.pop(key)
        elif l1 != l2:
            s1, s2 = int(l1 ** 0.5), int(l2 ** 0.5)
            resized = F.interpolate(src.permute(1, 0).reshape(1, h1, s1, s1).float(), size=(s2, s2), mode='bicubic')
            sd[key] = resized.view(h2, l2).permute(1, 0).to(src.dtype)
    for key in [k for k in sd if k.endswith('relative_position_index') and k in own and sd[k].shape != own[k].shape]:
        sd.pop(key)                                                          # a buffer derived from the window size: keep the model's
    return sd


def save_checkpoint(path, model, optimizer=None, meta=None):
    state = {'state_dict': model.state_dict(), 'meta': dict(meta or {})}
    if optimizer is not None:
        state['optimizer'] = optimizer.state_dict()
    tmp = path + '.tmp'
    torch.save(state, tmp)
    os.replace(tmp, path)
