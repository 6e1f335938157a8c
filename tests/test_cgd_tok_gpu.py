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


def _stage_operands(dev, dtype, shapes, seed=3):
    out = []
    for i, (B, P, C) in enumerate(shapes):
        s, t = _operands(B, P, C, seed + i, dtype)
        out.append((s.to(dev), t.to(dev)))
    return out


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_multi_job_call_equals_the_single_job_calls(dtype):
    """Several criteria in one call each way (ops.cgd_kl_tokens_multi: one scan launch per chunk class + ONE finish launch + ONE backward launch)
    give, bit for bit, what one call per criterion gives -- long chunks (64 pixels, 256 of them) and 16-pixel chunks in the same call,
    different widths, group sizes, temperatures, a permutation on two of them, a ragged tail."""
    from segdistill_amd import ops
    dev = torch.device('cuda:0')
    shapes = [(2, 16384, 64), (2, 1024, 768), (1, 333, 96), (2, 256, 768), (1, 4096, 128)]
    meta = [(8, 4.0, 3.0, None), (8, 4.0, 3.0, torch.randperm(768, generator=torch.Generator().manual_seed(1)).to(dev)), (7, 2.0, 1.0, None),
            (1, 1.0, 1.0, torch.randperm(768, generator=torch.Generator().manual_seed(2)).to(dev)), (128, 3.0, 2.0, None)]
    pairs = _stage_operands(dev, dtype, shapes)
    single = []
    for (s, t), (g, tau, alpha, perm) in zip(pairs, meta):
        s1 = s.clone().requires_grad_(True)
        loss, rows = ops.cgd_kl_tokens(s1, t, group_size=g, tau=tau, alpha=alpha, perm=perm, return_rows=True)
        (loss * 0.5).backward()
        single.append((loss.detach().clone(), rows.clone(), s1.grad.clone()))
    ss = [s.clone().requires_grad_(True) for s, _ in pairs]
    out = ops.cgd_kl_tokens_multi([(a, t) for a, (_, t) in zip(ss, pairs)], meta, return_rows=True)
    sum(l for l, _ in out).mul(0.5).backward()
    for (loss, rows), s_req, (l1, r1, g1) in zip(out, ss, single):
        assert torch.equal(loss, l1) and torch.equal(rows, r1) and torch.equal(s_req.grad, g1)
    # and against the oracle for the widest one
    B, P, C = shapes[1]
    ref = kd_ref.rowwise_kld(_as_nchw64(pairs[1][0].float().cpu()), _as_nchw64(pairs[1][1].float().cpu()), alpha=3.0, tau=4.0, perm=meta[1][3].cpu().numpy(),
                             group_size=8)
    assert float(out[1][0]) == pytest.approx(ref['loss'], rel=2e-5)


def test_finish_launch_is_run_to_run_identical_under_uneven_load():
    """The loss of a stage is written by whichever workgroup draws that stage's last arrival ticket (hand-off by agent-scope release /
    acquire): 30 launches of a four-stage call, with an unrelated streaming kernel racing on a second stream, must give the same bits, and
    the tickets must work again on every launch (they are re-zeroed by the scan launch)."""
    from segdistill_amd import ops
    dev = torch.device('cuda:0')
    shapes = [(8, 4096, 768), (8, 1024, 768), (8, 256, 768), (4, 16384, 256)]
    pairs = _stage_operands(dev, torch.bfloat16, shapes, seed=9)
    meta = [(8, 4.0, 3.0, None)] * 4
    first = None
    noise = torch.randn(64 << 20, device=dev)
    side = torch.cuda.Stream()
    for it in range(30):
        if it % 2:
            with torch.cuda.stream(side):
                noise.mul_(1.0001)
        out = ops.cgd_kl_tokens_multi(pairs, meta, return_rows=True)
        vals = torch.stack([l for l, _ in out]).clone()
        rows = torch.cat([r for _, r in out]).clone()
        if first is None:
            first = (vals, rows)
            for (s, t), l in zip(pairs, out):
                lone = ops.cgd_kl_tokens(s, t, group_size=8, tau=4.0, alpha=3.0)
                assert torch.equal(lone, l[0])
        else:
            assert torch.equal(vals, first[0]) and torch.equal(rows, first[1]), it
    torch.cuda.synchronize()
    # each loss IS the fixed-order sum of its rows
    for (l, r), (s, _) in zip(out, pairs):
        rows_n = s.shape[0] * (768 // 8 if s.shape[2] == 768 else 256 // 8)
        assert float(l) == pytest.approx(float(r.double().sum()) * 3.0 / rows_n, rel=1e-6)
