"""csrc/cgd_tok.hip: the CGD / CD criterion on token-major operands [B, P, C] against the fp64 oracle evaluated on the [B, C, h, w] view of the
same values (oracle/kd_ref.py:rowwise_kld, pinned to the reference's losses.py), and against the NCHW kernels (R1) on the transposed copy.
Covers: pad (C % g != 0), g = 1 / g = C / g > C, ragged pixel chunks, several vector positions per workgroup and more than 256 of them, a
channel permutation (shuffle iteration: the N channels of a lane's vector then belong to different rows), bf16 storage, an upstream factor."""
import numpy as np
import pytest
import torch

from oracle import kd_ref

pytestmark = pytest.mark.gpu

CASES = [  # B, P (pixels), C, g, tau, alpha
    (2, 64, 16, 8, 4.0, 3.0),
    (1, 1000, 24, 7, 2.0, 3.0),       # pad: 24 % 7, ragged chunks
    (2, 333, 768, 8, 4.0, 3.0),       # config-5 width: 192 fp32 / 96 bf16 vector positions
    (1, 4096, 32, 1, 1.0, 1.0),       # CD
    (1, 50, 40, 40, 3.0, 2.0),        # g = C
    (1, 77, 12, 32, 2.0, 1.0),        # g > C
    (1, 37, 1280, 10, 2.0, 3.0),      # more than 256 vector positions per pixel (fp32: 320)
    (3, 5, 8, 4, 1.0, 1.0),           # fewer pixels than one step
]


def _rel_l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.sqrt(((a - b) ** 2).sum() / max((b ** 2).sum(), 1e-300)))


def _operands(B, P, C, seed, dtype):
    g = torch.Generator().manual_seed(seed)
    s = (2 * torch.randn(B, P, C, generator=g)).to(dtype)
    t = (2 * torch.randn(B, P, C, generator=g)).to(dtype)
    return s, t


def _as_nchw64(x):
    b, p, c = x.shape
    return x.double().numpy().transpose(0, 2, 1).reshape(b, c, 1, p)


@pytest.mark.parametrize('case', CASES)
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('with_perm', [False, True])
def test_token_major_criterion_matches_oracle(case, dtype, with_perm):
    from segdistill_amd import ops
    B, P, C, g, tau, alpha = case
    n = 4 if dtype == torch.float32 else 8
    s, t = _operands(B, P, C, C * 31 + P, dtype)
    dev = torch.device('cuda:0')
    if C % n:
        assert not ops.cgd_kl_tokens_supported(s.to(dev), t.to(dev))
        return
    perm = torch.randperm(C, generator=torch.Generator().manual_seed(C)) if with_perm else None
    ref = kd_ref.rowwise_kld(_as_nchw64(s.float()), _as_nchw64(t.float()), alpha=alpha, tau=tau, perm=None if perm is None else perm.numpy(),
                             group_size=g)
    sg = s.to(dev).requires_grad_(True)
    assert ops.cgd_kl_tokens_supported(sg, t.to(dev), perm_len=None if perm is None else C)
    loss, rows = ops.cgd_kl_tokens(sg, t.to(dev), group_size=g, tau=tau, alpha=alpha, perm=None if perm is None else perm.to(dev), return_rows=True)
    up = 0.7
    (loss * up).backward()
    assert float(loss) == pytest.approx(ref['loss'], rel=2e-5, abs=1e-7)
    np.testing.assert_allclose(rows.cpu().numpy(), ref['row_kl'], rtol=3e-4, atol=2e-6)
    grad_ref = up * ref['grad_S'].reshape(B, C, P).transpose(0, 2, 1)
    tol = 1e-4 if dtype == torch.float32 else 6e-3           # the bf16 gradient is STORED in bf16
    assert _rel_l2(sg.grad.float().cpu().numpy(), grad_ref) < tol


def test_token_major_equals_nchw_kernels_at_config5_stage_shape():
    """[2, 64*64, 768] bf16 with a permutation: the token-major kernels and the R1 kernels on the transposed copy see the same rows."""
    from segdistill_amd import ops
    dev = torch.device('cuda:0')
    s, t = _operands(2, 4096, 768, 11, torch.bfloat16)
    perm = torch.randperm(768, generator=torch.Generator().manual_seed(5)).to(dev)
    s1 = s.to(dev).requires_grad_(True)
    l1 = ops.cgd_kl_tokens(s1, t.to(dev), group_size=8, tau=4.0, alpha=3.0, perm=perm)
    l1.backward()
    s2 = s.to(dev).transpose(1, 2).reshape(2, 768, 64, 64).contiguous().requires_grad_(True)
    l2 = ops.cgd_kl(s2, t.to(dev).transpose(1, 2).reshape(2, 768, 64, 64).contiguous(), group_size=8, tau=4.0, alpha=3.0, perm=perm)
    l2.backward()
    assert float(l1) == pytest.approx(float(l2), rel=1e-5)
    g2 = s2.grad.reshape(2, 768, 4096).transpose(1, 2)
    assert float((s1.grad.float() - g2.float()).norm() / g2.float().norm()) < 2e-3      # two bf16 roundings of the same fp32 values: identical up to ties
