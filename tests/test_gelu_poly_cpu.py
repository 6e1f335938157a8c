"""The branch-free erf form of the fused GELU epilogue (segdistill_amd/csrc/dwconv.hip::gelu_erf), restated in numpy float32 with the
coefficients READ FROM THE KERNEL SOURCE: the error bound documented there (and held on the GPU by
tests/test_dwconv_gpu.py::test_gelu_error_bound) must follow from the numbers actually compiled in."""
import os
import re

import numpy as np
from scipy.special import erfc

SRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'segdistill_amd', 'csrc', 'dwconv.hip')


def _kernel_constants():
    with open(SRC) as f:
        body = f.read()
    body = body[body.index('float gelu_erf(float v)'):]
    body = body[:body.index('\n}')]
    rs2 = float(re.search(r'fabsf\(v\) \* ([0-9.]+)f', body).group(1))
    p = float(re.search(r'fmaf\(([0-9.]+)f, a, 1\.f\)', body).group(1))
    lead = float(re.search(r'float p = (-?[0-9.]+)f;', body).group(1))
    rest = [float(x) for x in re.findall(r'p = fmaf\(p, t, (-?[0-9.]+)f\);', body)]
    log2e = float(re.search(r'a \* a \* (-?[0-9.]+)f', body).group(1))
    return rs2, p, [lead] + rest, log2e


def test_kernel_constants_are_the_documented_fit():
    rs2, p, coef, log2e = _kernel_constants()
    assert abs(rs2 - 2 ** -0.5) < 1e-12 and abs(log2e + 1.0 / np.log(2.0)) < 1e-12
    assert p == 0.39 and len(coef) == 6
    # exact-arithmetic error of erfc(a) = t P(t) exp(-a^2) on [0, 6]
    a = np.linspace(0.0, 6.0, 120001)
    t = 1.0 / (1.0 + p * a)
    poly = np.zeros_like(a)
    for c in coef:
        poly = poly * t + c
    err = np.abs(poly * t * np.exp(-a * a) - erfc(a))
    assert err.max() < 1.2e-8          # 1.1e-8 in the kernel's comment


def test_float32_evaluation_meets_the_documented_bound():
    rs2, p, coef, log2e = _kernel_constants()
    f = np.float32
    v = np.linspace(-12, 12, 2000001).astype(f)
    a = np.abs(v) * f(rs2)
    t = f(1) / (f(1) + f(p) * a)
    poly = np.full_like(a, f(coef[0]))
    for c in coef[1:]:
        poly = poly * t + f(c)
    q = poly * t * np.exp2(a * a * f(log2e)).astype(f)
    g = (f(0.5) * v) * np.where(v < 0, q, f(2) - q)
    x = v.astype(np.float64)
    ref = 0.5 * x * erfc(-x / np.sqrt(2.0))
    err = np.abs(g - ref)
    assert err.max() < 6e-7                                           # the GPU test's bound
    tail = (x < -3) & (x > -9)
    assert (err / np.abs(ref))[tail].max() < 2e-2                     # relative accuracy kept where 1 + erf cancels
    mid = np.abs(x) < 3
    assert (err / np.maximum(np.abs(ref), 1e-6))[mid].max() < 1e-5
