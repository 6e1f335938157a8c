"""Image primitives with OpenCV semantics in numpy (mmcv.imresize / imrescale / imflip / impad / imnormalize, bgr2hsv / hsv2bgr)."""
from __future__ import annotations

import numpy as np


def rescale_size(old_wh, scale):
    """mmcv.rescale_size: `scale` a float factor, or (long_edge_max, short_edge_max)."""
    w, h = old_wh
    if isinstance(scale, (float, int)):
        if scale <= 0:
            raise ValueError(f'Invalid scale {scale}, must be positive.')
        f = scale
    else:
        long_e, short_e = max(scale), min(scale)
        f = min(long_e / max(h, w), short_e / min(h, w))
    return int(w * float(f) + 0.5), int(h * float(f) + 0.5)


def imresize(img, size_wh, interpolation='bilinear'):
    """cv2.resize(img, (w, h)): INTER_LINEAR (half-pixel centres, edge clamp, no antialias) or INTER_NEAREST (floor rule)."""
    W, H = int(size_wh[0]), int(size_wh[1])
    h, w = img.shape[:2]
    if (w, h) == (W, H):
        return img.copy()
    if interpolation == 'nearest':
        ys = np.minimum((np.arange(H) * (h / H)).astype(np.int64), h - 1)
        xs = np.minimum((np.arange(W) * (w / W)).astype(np.int64), w - 1)
        return img[ys][:, xs]
    if interpolation != 'bilinear':
        raise ValueError(f'unsupported interpolation {interpolation!r}')

    def taps(n_out, n_in):
        s = (np.arange(n_out, dtype=np.float64) + 0.5) * (n_in / n_out) - 0.5
        i0 = np.floor(s).astype(np.int64)
        lam = s - i0
        lam = np.where(i0 < 0, 0.0, lam)            # cv2: sx < 0 -> fx = 0, sx = 0
        i0 = np.clip(i0, 0, n_in - 1)
        i1 = np.clip(i0 + 1, 0, n_in - 1)
        return i0, i1, lam

    y0, y1, ly = taps(H, h)
    x0, x1, lx = taps(W, w)
    src = img.astype(np.float64)
    if src.ndim == 2:
        src = src[:, :, None]
    top = src[y0][:, x0] * (1 - lx)[None, :, None] + src[y0][:, x1] * lx[None, :, None]
    bot = src[y1][:, x0] * (1 - lx)[None, :, None] + src[y1][:, x1] * lx[None, :, None]
    out = top * (1 - ly)[:, None, None] + bot * ly[:, None, None]
    if img.dtype == np.uint8:
        out = np.clip(np.rint(out), 0, 255).astype(np.uint8)
    else:
        out = out.astype(img.dtype)
    return out[:, :, 0] if img.ndim == 2 else out


def imrescale(img, scale, interpolation='bilinear'):
    h, w = img.shape[:2]
    return imresize(img, rescale_size((w, h), scale), interpolation)


def imflip(img, direction='horizontal'):
    if direction == 'horizontal':
        return np.flip(img, axis=1)
    if direction == 'vertical':
        return np.flip(img, axis=0)
    raise ValueError(direction)


def impad(img, shape_hw, pad_val=0):
    H, W = shape_hw
    h, w = img.shape[:2]
    if H < h or W < w:
        raise ValueError(f'pad shape {shape_hw} smaller than the image {(h, w)}')
    out = np.full((H, W) + img.shape[2:], pad_val, dtype=img.dtype)
    out[:h, :w] = img
    return out


def imnormalize(img, mean, std, to_rgb=True):
    img = img.astype(np.float32)
    if to_rgb:
        img = img[..., ::-1]
    return (img - np.asarray(mean, np.float32).reshape(1, 1, -1)) / np.asarray(std, np.float32).reshape(1, 1, -1)


def bgr2hsv(img):
    """8-bit cv2.COLOR_BGR2HSV: H in [0,180), S and V in [0,255]."""
    f = img.astype(np.float64)
    b, g, r = f[..., 0], f[..., 1], f[..., 2]
    v = np.maximum(np.maximum(b, g), r)
    mn = np.minimum(np.minimum(b, g), r)
    d = v - mn
    s = np.where(v > 0, d / np.where(v > 0, v, 1) * 255.0, 0.0)
    safe = np.where(d > 0, d, 1)
    hh = np.where(v == r, (g - b) / safe, np.where(v == g, 2.0 + (b - r) / safe, 4.0 + (r - g) / safe)) * 60.0
    hh = np.where(d > 0, hh, 0.0)
    hh = np.where(hh < 0, hh + 360.0, hh) / 2.0
    h8 = np.rint(hh).astype(np.int64) % 180
    return np.stack([h8, np.clip(np.rint(s), 0, 255), v], axis=-1).astype(np.uint8)


def hsv2bgr(img):
    """8-bit cv2.COLOR_HSV2BGR."""
    f = img.astype(np.float64)
    h, s, v = f[..., 0] * 2.0, f[..., 1] / 255.0, f[..., 2]
    h = np.where(h >= 360.0, h - 360.0, h) / 60.0
    sector = np.floor(h).astype(np.int64) % 6
    fr = h - np.floor(h)
    p, q, t = v * (1 - s), v * (1 - s * fr), v * (1 - s * (1 - fr))
    r = np.choose(sector, [v, q, p, p, t, v])
    g = np.choose(sector, [t, v, v, q, p, p])
    b = np.choose(sector, [p, p, t, v, v, q])
    return np.clip(np.rint(np.stack([b, g, r], axis=-1)), 0, 255).astype(np.uint8)
