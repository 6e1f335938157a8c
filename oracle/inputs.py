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
