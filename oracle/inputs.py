"""Deterministic, RNG-free synthetic operands shared by the golden generator and
the tests (TEST INFRASTRUCTURE).  Closed-form so large cases need not be stored."""
from __future__ import annotations

import numpy as np


def kat_pair(shape=(2, 6, 4, 4)):
    """G1 known-answer operands: s=2 sin(0.1 k), t=2 cos(0.13 k),
    k = 7b+3c+5h+11w, fp64."""
    b, c, h, w = np.meshgrid(*[np.arange(n) for n in shape], indexing='ij')
    k = 7 * b + 3 * c + 5 * h + 11 * w
    return 2 * np.sin(0.1 * k), 2 * np.cos(0.13 * k)


def wavy_pair(shape, scale=2.0, dtype=np.float32):
    """Logit-like pseudo-random operands from incommensurate phases; returned in
    ``dtype`` (fp32 by default so fp32 and fp64 consumers see identical values)."""
    n = int(np.prod(shape))
    i = np.arange(n, dtype=np.float64)
    s = scale * np.sin(i * 0.7371 + 2.3 * np.sin(i * 0.01113)) + 0.5 * scale * np.cos(i * 0.113)
    t = scale * np.cos(i * 0.5917 + 1.7 * np.cos(i * 0.00731)) + 0.5 * scale * np.sin(i * 0.271 + 0.4)
    return s.reshape(shape).astype(dtype), t.reshape(shape).astype(dtype)


def probe_vector(shape):
    """Fixed direction used to compress a gradient into one number: <grad, probe>."""
    n = int(np.prod(shape))
    return np.cos(0.37 * np.arange(n, dtype=np.float64) + 0.1).reshape(shape)


def fill_state_dict_(module, salt=0):
    """Overwrite every parameter / float buffer of a torch module with closed-form values that
    depend only on the tensor's NAME and shape, so that two implementations with identical
    state-dict keys (the reference's and ours) hold identical weights without storing them."""
    import zlib

    import torch
    sd = module.state_dict()
    with torch.no_grad():
        for name, t in sd.items():
            if not torch.is_floating_point(t):
                continue
            n = t.numel()
            phase = (zlib.crc32(name.encode()) % 10007) / 10007.0 * 6.283185307179586 + 0.7391 * salt
            base = torch.sin(torch.arange(n, dtype=torch.float64) * 0.6180339887 + phase)
            if name.endswith('running_var'):
                v = 1.0 + 0.2 * base * base
            elif name.endswith('running_mean'):
                v = 0.1 * base
            elif t.dim() <= 1 and name.endswith('weight'):
                v = 1.0 + 0.1 * base
            elif t.dim() <= 1:
                v = 0.05 * base
            elif 'relative_position_bias_table' in name or 'absolute_pos_embed' in name:
                v = 0.2 * base
            else:
                fan_in = n // t.shape[0]
                v = base * (1.5 / max(fan_in, 1) ** 0.5)
            t.copy_(v.reshape(t.shape).to(t.dtype))
    return module


def splitmix_uniform(n, seed):
    """n reproducible uniforms in [0, 1): the splitmix64 finaliser applied to the counter seed+i.  Integer arithmetic only
    (wrap-around uint64), so every platform produces the same bits -- unlike a sine-based hash, whose last-ulp differences
    would be amplified into different values."""
    z = (np.arange(n, dtype=np.uint64) + np.uint64(seed % (1 << 64))) * np.uint64(0x9E3779B97F4A7C15)
    z ^= z >> np.uint64(30)
    z *= np.uint64(0xBF58476D1CE4E5B9)
    z ^= z >> np.uint64(27)
    z *= np.uint64(0x94D049BB133111EB)
    z ^= z >> np.uint64(31)
    return (z >> np.uint64(11)).astype(np.float64) / float(1 << 53)


def fill_state_dict_hashed_(module, salt=0, gain=1.4):
    """Like fill_state_dict_, with decorrelated (hash-uniform) values: matrices / filters ~ U(-a, a), a = sqrt(3) gain / sqrt(fan_in)
    (unit-gain He-like variance).  The smooth sine weights of fill_state_dict_ give strongly correlated features, tiny batch
    variances and gradients that the reference's OWN fp32 run reproduces only to 1-90 %; with these the reference's fp32 run
    agrees with its fp64 run to ~1e-5 on every probe (recorded per value as '@fp32dev'), so fp32 consumers can hold 1e-3."""
    import zlib

    import torch
    sd = module.state_dict()
    with torch.no_grad():
        for name, t in sd.items():
            if not torch.is_floating_point(t):
                continue
            n = t.numel()
            u = 2.0 * splitmix_uniform(n, (zlib.crc32(name.encode()) + 7919 * salt) * 1000003) - 1.0
            if name.endswith('running_var'):
                v = 1.0 + 0.2 * u * u
            elif name.endswith('running_mean'):
                v = 0.1 * u
            elif t.dim() <= 1 and name.endswith('weight'):
                v = 1.0 + 0.1 * u
            elif t.dim() <= 1:
                v = 0.05 * u
            elif 'relative_position_bias_table' in name or 'absolute_pos_embed' in name:
                v = 0.2 * u
            else:
                fan_in = n // t.shape[0]
                v = u * (3 ** 0.5) * gain / max(fan_in, 1) ** 0.5
            t.copy_(torch.from_numpy(v.reshape(tuple(t.shape))).to(t.dtype))
    return module


def wavy_image(shape=(2, 3, 64, 64)):
    """Synthetic normalised image batch (float32) and a label map with some 255 (ignore) pixels."""
    n = int(np.prod(shape))
    i = np.arange(n, dtype=np.float64)
    img = (np.sin(i * 0.01931 + 0.5 * np.cos(i * 0.000371)) + 0.3 * np.cos(i * 0.7713)).reshape(shape).astype(np.float32)
    b, _, h, w = shape
    yy, xx = np.meshgrid(np.arange(h), np.arange(w), indexing='ij')
    lab = np.stack([((yy // 7 + 2 * (xx // 5) + 3 * k) % 150) for k in range(b)]).astype(np.int64)
    lab[:, ::9, ::4] = 255
    return img, lab[:, None]
