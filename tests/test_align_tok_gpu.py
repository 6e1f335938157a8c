"""csrc/align_tok.hip: the feature-align projection of a token-major bf16 tap fused with the channel-group criterion (SURVEY a-15 + a-16; reference
opts.py:25-27 docstring, commented `self.ff` of losses.py:258,332-333,373-374, KLDLoss.forward losses.py:95-113) against the fp64 oracle

    y = x . W_bf16^T + b  (fp64 on the bf16-rounded operands; NOT rounded: the fused kernels never store it)  ->  oracle/kd_ref.rowwise_kld

Covers: K in {64, 128, 256}; pad (C % g != 0); a channel permutation; ragged token tiles (P % 512, P % 64, P < 64); several images; several jobs of
different size in one call (the plan's channel split of the small jobs); an upstream factor; no bias; the stored dY itself (bf16 of the fp64 value);
the stand-alone projection (plain mode) against torch."""
import numpy as np
import pytest
import torch

from oracle import kd_ref

pytestmark = pytest.mark.gpu

CASES = [  # B, P, K, C, g, tau, alpha, perm, bias
    (1, 256, 256, 768, 8, 4.0, 3.0, False, True),      # one full tile at config 5's widths
    (2, 300, 256, 96, 8, 4.0, 3.0, True, True),        # a ragged tile (300 rows: four full waves, one of 44 rows, three empty)
    (1, 40, 64, 32, 1, 1.0, 1.0, False, True),         # fewer rows than one wave; CD rows (g = 1)
    (3, 513, 128, 160, 7, 2.0, 2.0, True, False),      # pad 160 % 7, a one-row second tile, no bias
    (1, 1024, 256, 768, 8, 4.0, 3.0, True, True),      # config 5 stage 3, B = 1
    (2, 64, 64, 64, 64, 3.0, 1.0, False, True),        # g = C
]


def _rel_l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.sqrt(((a - b) ** 2).sum() / max((b ** 2).sum(), 1e-300)))


def _operands(B, P, K, C, seed, bias):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, P, K, generator=g).to(torch.bfloat16)
    w = (torch.randn(C, K, generator=g) * (2.0 / K ** 0.5))
    b = (0.3 * torch.randn(C, generator=g)) if bias else None
    t = (2 * torch.randn(B, P, C, generator=g)).to(torch.bfloat16)
    return x, w, b, t


def _oracle(x, w, b, t, g, tau, alpha, perm):
    B, P, K = x.shape
    C = w.shape[0]
    x64 = x.double().numpy()
    w64 = w.to(torch.bfloat16).double().numpy()                    # the kernels read the bf16 copy of the fp32 master weight
    y = x64 @ w64.T + (0 if b is None else b.double().numpy())
    t64 = t.double().numpy()
    nchw = lambda a: a.transpose(0, 2, 1).reshape(B, a.shape[2], 1, P)
    ref = kd_ref.rowwise_kld(nchw(y), nchw(t64), alpha=alpha, tau=tau, perm=None if perm is None else perm.numpy(), group_size=g)
    dy = ref['grad_S'].reshape(B, C, P).transpose(0, 2, 1)          # [B, P, C]
    return {'loss': ref['loss'], 'row_kl': ref['row_kl'], 'dy': dy, 'dx': dy @ w64, 'dw': np.einsum('bpc,bpk->ck', dy, x64), 'db': dy.sum((0, 1))}


@pytest.mark.parametrize('case', CASES)
def test_fused_align_criterion_matches_oracle(case):
    from segdistill_amd import ops
    B, P, K, C, g, tau, alpha, with_perm, bias = case
    dev = torch.device('cuda:0')
    x, w, b, t = _operands(B, P, K, C, 7 * C + P, bias)
    perm = torch.randperm(C, generator=torch.Generator().manual_seed(C)) if with_perm else None
    ref = _oracle(x, w, b, t, g, tau, alpha, perm)
    xg = x.to(dev).requires_grad_(True)
    wg = w.to(dev).requires_grad_(True)
    bg = None if b is None else b.to(dev).requires_grad_(True)
    assert ops.align_cgd_tokens_supported(xg, wg, t.to(dev))
    (loss, rows), = ops.align_cgd_tokens_multi([(xg, wg, bg, t.to(dev))], [(g, tau, alpha, None if perm is None else perm.to(dev))], defer_ok=False,
                                               return_rows=True)
    up = 0.7
    (loss * up).backward()
    torch.cuda.synchronize()
    assert float(loss) == pytest.approx(ref['loss'], rel=3e-5, abs=1e-7)
    np.testing.assert_allclose(rows.cpu().numpy(), ref['row_kl'], rtol=5e-4, atol=3e-6)
    # gradients pass through a bf16-stored dY (8 significant bits, rounded once)
    assert _rel_l2(xg.grad.float().cpu().numpy(), up * ref['dx']) < 6e-3
    assert _rel_l2(wg.grad.cpu().numpy(), up * ref['dw']) < 6e-3
    if bias:
        # the column sums are taken before the rounding; measured against the mass they sum (with g = 1 every column sums to zero exactly)
        mass = up * np.abs(ref['dy']).sum((0, 1))
        assert (np.abs(bg.grad.cpu().numpy() - up * ref['db']) <= 2e-5 * mass + 1e-12).all()


def test_stored_dy_is_the_bf16_of_the_oracle_gradient():
    """The backward launch through the C ABI: dY [B, P, C] elementwise against the fp64 gradient (one bf16 rounding + fp32 arithmetic), the
    per-tile column sums against the sums of the fp64 gradient over each 512-token tile."""
    import ctypes as C
    from segdistill_amd import _lib, ops
    B, P, K, Cc, g, tau, alpha = 2, 600, 256, 128, 8, 4.0, 3.0
    dev = torch.device('cuda:0')
    x, w, b, t = _operands(B, P, K, Cc, 3, True)
    ref = _oracle(x, w, b, t, g, tau, alpha, None)
    xg = x.to(dev).requires_grad_(True)
    wg, bg, tg = w.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True), t.to(dev)
    (loss, _), = ops.align_cgd_tokens_multi([(xg, wg, bg, tg)], [(g, tau, alpha, None)], defer_ok=False, return_rows=True)
    fn = loss.grad_fn
    row_lse2 = fn.saved_tensors[4]
    wc = fn.saved_tensors[1]
    L = _lib.lib()
    tiles = L.sd_align_cgd_tok_tiles(B, P)
    per_img = L.sd_align_cgd_tok_tiles(1, P)
    tile_rows = 512                                                  # tokens per workgroup item (csrc/align_tok.hip: kBM)
    assert tiles == B * per_img and per_img == -(-P // tile_rows)
    dY = torch.full((B, P, Cc), float('nan'), dtype=torch.bfloat16, device=dev)
    dbp = torch.empty(tiles, Cc, dtype=torch.float32, device=dev)
    job = (ops._AlignTokJob * 1)()
    rows = B * (Cc // g)
    j = job[0]
    j.X, j.W, j.bias, j.T, j.row_lse2, j.out, j.db_part = xg.data_ptr(), wc.data_ptr(), bg.data_ptr(), tg.data_ptr(), row_lse2.data_ptr(), dY.data_ptr(), dbp.data_ptr()
    j.P, j.B, j.K, j.C, j.g, j.inv_tau, j.coef = P, B, K, Cc, g, 1.0 / tau, alpha / (rows * tau)
    _lib.check(L.sd_align_cgd_tok_bwd_multi(C.cast(job, C.c_void_p), 1, torch.cuda.current_stream().cuda_stream), 'bwd')
    torch.cuda.synchronize()
    got = dY.float().cpu().numpy().astype(np.float64)
    assert np.isfinite(got).all()
    err = np.abs(got - ref['dy'])
    assert (err <= 2.0 ** -8 * np.abs(ref['dy']) + 1e-3 * np.abs(ref['dy']).max()).all()
    part = np.zeros((tiles, Cc))
    for bi in range(B):
        for kb in range(per_img):
            part[bi * per_img + kb] = ref['dy'][bi, kb * tile_rows:(kb + 1) * tile_rows].sum(0)
    np.testing.assert_allclose(dbp.cpu().numpy(), part, rtol=2e-3, atol=2e-4 * np.abs(part).max())


def test_several_jobs_in_one_call_match_single_calls():
    """Config 5's four stages at B = 1 (32 + 8 + 2 + 1 tiles: the plan cuts every item into channel ranges) against one call per stage."""
    from segdistill_amd import ops
    dev = torch.device('cuda:0')
    K, C = 256, 768
    stages = [16384, 4096, 1024, 256]
    g0 = torch.Generator().manual_seed(1)
    w = [(torch.randn(C, K, generator=g0) / 16).to(dev).requires_grad_(True) for _ in stages]
    b = [(0.1 * torch.randn(C, generator=g0)).to(dev).requires_grad_(True) for _ in stages]
    xs = [torch.randn(1, P, K, generator=g0).to(torch.bfloat16).to(dev).requires_grad_(True) for P in stages]
    ts = [(2 * torch.randn(1, P, C, generator=g0)).to(torch.bfloat16).to(dev) for P in stages]
    perm = torch.randperm(C, generator=g0).to(dev)
    meta = [(8, 4.0, 3.0, perm)] * 4
    multi = ops.align_cgd_tokens_multi(list(zip(xs, w, b, ts)), meta, defer_ok=False)
    sum(multi).backward()
    got = [(float(l), x.grad.clone(), ww.grad.clone(), bb.grad.clone()) for l, x, ww, bb in zip(multi, xs, w, b)]
    for v in xs + w + b:
        v.grad = None
    for i in range(4):
        l, = ops.align_cgd_tokens_multi([(xs[i], w[i], b[i], ts[i])], [meta[i]], defer_ok=False)
        l.backward()
        assert float(l) == pytest.approx(got[i][0], rel=1e-6)
        assert torch.equal(xs[i].grad, got[i][1])                 # same tiles, same arithmetic: bit-identical whatever the channel split
        assert _rel_l2(w[i].grad.cpu().numpy(), got[i][2].cpu().numpy()) < 1e-6
        assert _rel_l2(b[i].grad.cpu().numpy(), got[i][3].cpu().numpy()) < 1e-6


def test_fused_entry_equals_the_unfused_product_path():
    """DistillationLoss.entry_loss with SEGDISTILL_ALIGN_FUSED on / off on the same taps: the unfused path stores the projected feature in bf16,
    so the two agree to that rounding (loss 1e-3, gradients 2e-2)."""
    from segdistill_amd import ops
    from segdistill_amd.distillation.opts import DistillationLoss
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    dl = DistillationLoss([dict(student_layer='a', teacher_layer='a', loss_name='KLDLoss', channel_nums=(256, 768),
                                loss_config=dict(alpha=3, tau=4, transform_config={'loss_type': 'channel', 'group_size': 8}))]).to(dev)
    with torch.no_grad():
        dl.aligns['0'].bias.copy_(0.05 * torch.sin(torch.arange(768.0)))
    x = torch.randn(2, 1024, 256, device=dev).to(torch.bfloat16)
    t = (2 * torch.randn(2, 1024, 768, device=dev)).to(torch.bfloat16)
    res = []
    for fused in (True, False):
        ops._ALIGN_FUSED = fused
        try:
            xg = x.clone().requires_grad_(True)
            dl.aligns['0'].weight.grad = dl.aligns['0'].bias.grad = None
            loss = dl.entry_loss(0, xg, t, None, 1)
            loss.backward()
            res.append((float(loss), xg.grad.float().cpu().numpy(), dl.aligns['0'].weight.grad.cpu().numpy(), dl.aligns['0'].bias.grad.cpu().numpy()))
        finally:
            ops._ALIGN_FUSED = True
    assert res[0][0] == pytest.approx(res[1][0], rel=1e-3)
    for k in (1, 2, 3):
        assert _rel_l2(res[0][k], res[1][k]) < 2e-2


@pytest.mark.parametrize('shape', [(1000, 256, 768), (256, 64, 32), (70, 128, 96)])
def test_plain_projection_matches_torch(shape):
    from segdistill_amd import _lib
    T, K, C = shape
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(T)
    x = torch.randn(T, K, generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn(C, K, generator=g) / K ** 0.5).to(torch.bfloat16).to(dev)
    b = torch.randn(C, generator=g).to(dev)
    y = torch.full((T, C), float('nan'), dtype=torch.bfloat16, device=dev)
    _lib.check(_lib.lib().sd_linear_tok_bf16_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), T, K, C, torch.cuda.current_stream().cuda_stream), 'plain')
    ref = (x.double() @ w.double().t() + b.double()).cpu().numpy()
    got = y.double().cpu().numpy()
    assert np.isfinite(got).all()
    assert (np.abs(got - ref) <= 2.0 ** -8 * np.abs(ref) + 1e-4).all()


@pytest.mark.parametrize('shape', [(1000, 768, 256), (256, 32, 64), (70, 96, 128), (4096, 768, 256)])
def test_input_gradient_gemm_matches_torch(shape):
    """dX = dY . W (csrc/align_tok.hip: tok_dx_kernel) against fp64 on the same bf16 operands; ragged token tiles, all three student widths."""
    from segdistill_amd import _lib
    T, C, N = shape
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(T + C)
    dy = torch.randn(T, C, generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn(C, N, generator=g) / C ** 0.5).to(torch.bfloat16).to(dev)
    dx = torch.full((T, N), float('nan'), dtype=torch.bfloat16, device=dev)
    _lib.check(_lib.lib().sd_linear_tok_bf16_bwd_data(dy.data_ptr(), w.data_ptr(), dx.data_ptr(), T, C, N, torch.cuda.current_stream().cuda_stream), 'bwd_data')
    ref = (dy.double() @ w.double()).cpu().numpy()
    got = dx.double().cpu().numpy()
    assert np.isfinite(got).all()
    assert (np.abs(got - ref) <= 2.0 ** -8 * np.abs(ref) + 2e-4).all()
