"""The arithmetic claim behind the split-bf16 ("bf16x3") kernels (csrc/token_gemm.hip, csrc/sra_attn.hip), checked in numpy on the CPU:
  * a float32 splits EXACTLY into three bf16 terms by two round-to-nearest residual steps (8 + 8 + 8 significand bits);
  * a product formed from the six cross terms the kernels keep (hh, hm, mh, hl, lh, mm) differs from the exact product by at most
    ~2^-22 relative -- the rounding level of an fp32 multiply-add chain -- so a length-K dot product accumulated in fp32 is as accurate
    as the f32-input MFMA path.
The GPU tests hold the kernels themselves to these bounds (tests/test_token_gemm_gpu.py, tests/test_sra_gpu.py::test_split_bf16_error_bound)."""
import numpy as np


def bf16_rn(x):
    """float32 -> nearest bf16 (ties to even), returned as float32 (what v_cvt_pk_bf16_f32 does for finite values)."""
    u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def split3(x):
    hi = bf16_rn(x)
    r1 = (x - hi).astype(np.float32)
    mid = bf16_rn(r1)
    r2 = (r1 - mid).astype(np.float32)
    lo = bf16_rn(r2)
    return hi, mid, lo, r2


def _samples(n, seed):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal(n).astype(np.float32) * np.exp2(rng.integers(-30, 30, n)).astype(np.float32)
    edge = np.array([1.0, -1.0, 1.0 + 2 ** -23, 1.0 - 2 ** -24, 3.0e38, -3.0e38, 1.1754944e-38, 0.0, 255.99998, 2 ** -100], dtype=np.float32)
    return np.concatenate([x, edge])


def test_three_bf16_terms_reproduce_a_float32_exactly():
    x = _samples(200000, 1)
    hi, mid, lo, r2 = split3(x)
    assert np.array_equal(lo, r2)                                     # the second residual already is a bf16: nothing is lost
    total = hi.astype(np.float64) + mid.astype(np.float64) + lo.astype(np.float64)
    assert np.array_equal(total, x.astype(np.float64))
    # every residual of a round-to-nearest step is exactly representable in float32 (so `x - hi` itself is exact)
    assert np.array_equal((x.astype(np.float64) - hi.astype(np.float64)).astype(np.float32).astype(np.float64), x.astype(np.float64) - hi.astype(np.float64))


def test_six_cross_terms_are_fp32_grade():
    a, b = _samples(100000, 2)[:100000], _samples(100000, 3)[:100000]
    ah, am, al, _ = (t.astype(np.float64) for t in split3(a))
    bh, bm, bl, _ = (t.astype(np.float64) for t in split3(b))
    kept = am * bm + al * bh + ah * bl + am * bh + ah * bm + ah * bh   # the kernels' order: small terms first
    exact = a.astype(np.float64) * b.astype(np.float64)
    nz = exact != 0
    rel = np.abs(kept - exact)[nz] / np.abs(exact)[nz]
    assert rel.max() < 2.0 ** -22                                     # dropped: mid.lo + lo.mid + lo.lo <= 2 * 2^-8 * 2^-16 (+ 2^-32)


def test_dot_product_error_matches_an_fp32_fma_chain():
    rng = np.random.default_rng(4)
    K = 512
    A = rng.standard_normal((64, K)).astype(np.float32)
    B = rng.standard_normal((K, 64)).astype(np.float32)
    exact = A.astype(np.float64) @ B.astype(np.float64)
    ah, am, al, _ = split3(A)
    bh, bm, bl, _ = split3(B)
    acc = np.zeros((64, 64), dtype=np.float32)
    for k0 in range(0, K, 16):                                        # one MFMA = 16 k: products exact, fp32 accumulation per instruction
        s = slice(k0, k0 + 16)
        for x, y in ((am, bm), (al, bh), (ah, bl), (am, bh), (ah, bm), (ah, bh)):
            acc = (acc.astype(np.float64) + x[:, s].astype(np.float64) @ y[s, :].astype(np.float64)).astype(np.float32)
    ref32 = np.zeros((64, 64), dtype=np.float32)
    for k in range(K):                                                # the f32-input MFMA path: an fmaf chain over k
        ref32 = (ref32.astype(np.float64) + np.outer(A[:, k], B[k, :]).astype(np.float64)).astype(np.float32)
    scale = np.abs(exact).max()
    e_split, e_f32 = np.abs(acc - exact).max() / scale, np.abs(ref32 - exact).max() / scale
    assert e_split < 2e-6 and e_split < 4 * e_f32 + 1e-7              # the bound the GPU tests use (2e-6), and the same order as fp32 itself
