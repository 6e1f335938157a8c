"""csrc/token_gemm.hip: the token-major Linear forward (+ bias / exact GELU / residual) and input gradient on the f32-input MFMA, against
fp64 torch arithmetic.  The MFMA result is bit-equal to an fp32 fmaf chain, so the error budget is that of a K-term fp32 dot product
(~1e-7 sqrt(K) relative to |x|.|w|); shapes cover every tile configuration (N <= 32, <= 64, > 64), ragged M / N / K edges, the unaligned
(scalar-load) path (K or N not a multiple of 4: linear_pred's 150 classes), single- and many-step K loops, and the asymmetric-identity check
that catches a transposed accumulator layout."""
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [  # tokens, in_features (K), out_features (N)
    (4096, 32, 32), (1000, 32, 128), (4099, 64, 64), (2048, 128, 32), (777, 160, 640), (512, 640, 160), (300, 256, 150), (260, 150, 256),
    (129, 36, 40), (64, 1024, 64), (2048, 512, 2048), (31, 7, 5), (16384, 64, 256),
]


def _rel(a, b):
    return float((a.double() - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize('T,K,N', SHAPES)
def test_linear_fwd_and_bwd_data_match_fp64(T, K, N):
    from segdistill_amd import token_gemm
    dev = torch.device('cuda:0')
    g = torch.Generator(device=dev).manual_seed(T * 7 + K * 3 + N)
    x = torch.randn(T, K, device=dev, generator=g)
    w = torch.randn(N, K, device=dev, generator=g) / K ** 0.5
    b = torch.randn(N, device=dev, generator=g)
    r = torch.randn(T, N, device=dev, generator=g)
    dy = torch.randn(T, N, device=dev, generator=g)
    x64, w64, b64 = x.double(), w.double(), b.double()
    ref = x64 @ w64.t() + b64
    tol = 3e-7 * max(8.0, K ** 0.5)
    assert _rel(token_gemm.linear_fwd(x, w, b), ref) < tol
    assert _rel(token_gemm.linear_fwd(x, w), x64 @ w64.t()) < tol
    assert _rel(token_gemm.linear_fwd(x, w, b, act='gelu'), torch.nn.functional.gelu(ref)) < tol + 2e-7
    assert _rel(token_gemm.linear_fwd(x, w, b, residual=r), ref + r.double()) < tol
    assert _rel(token_gemm.linear_bwd_data(dy, w), dy.double() @ w64) < 3e-7 * max(8.0, N ** 0.5)
    # split-bf16 arithmetic: the SAME bound as the f32-input MFMA path, and its error measured next to that path's
    y3, y1 = token_gemm.linear_fwd(x, w, b, split_bf16=True), token_gemm.linear_fwd(x, w, b)
    assert _rel(y3, ref) < tol
    assert _rel(token_gemm.linear_fwd(x, w, b, residual=r, split_bf16=True), ref + r.double()) < tol
    assert _rel(token_gemm.linear_bwd_data(dy, w, split_bf16=True), dy.double() @ w64) < 3e-7 * max(8.0, N ** 0.5)
    e3, e1 = float((y3.double() - ref).norm() / ref.norm()), float((y1.double() - ref).norm() / ref.norm())
    assert e3 < 2.0 * e1 + 1e-8, (e3, e1)          # rel-L2 error of the split path within 2x of the exact-f32 path


def test_asymmetric_identity_and_3d_input():
    """W = [I | 0] with an asymmetric X: a swapped row/column map in the accumulator write or the operand fragments cannot pass."""
    from segdistill_amd import token_gemm
    dev = torch.device('cuda:0')
    T, K, N = 192, 96, 80
    x = (torch.arange(T * K, device=dev, dtype=torch.float32).reshape(2, T // 2, K) % 251) - 100.0     # exact small integers
    w = torch.zeros(N, K, device=dev)
    w[torch.arange(N), torch.arange(N)] = 1.0
    y = token_gemm.linear_fwd(x, w)
    assert y.shape == (2, T // 2, N) and torch.equal(y, x[..., :N])
    dy = (torch.arange(T * N, device=dev, dtype=torch.float32).reshape(2, T // 2, N) % 127) - 60.0
    dx = token_gemm.linear_bwd_data(dy, w)
    assert torch.equal(dx[..., :N], dy) and float(dx[..., N:].abs().max()) == 0.0


def test_row_strided_weight_view_is_taken_without_a_copy():
    """The per-branch [E, E] blocks of the SegFormer head's linear_fuse weight [E, 4E]: unit column stride, rows 4E apart."""
    from segdistill_amd import token_gemm
    dev = torch.device('cuda:0')
    g = torch.Generator(device=dev).manual_seed(3)
    big = torch.randn(96, 4 * 96, device=dev, generator=g) / 10
    blk = big[:, 96:192]
    x = torch.randn(3000, 96, device=dev, generator=g)
    dy = torch.randn(3000, 96, device=dev, generator=g)
    assert _rel(token_gemm.linear_fwd(x, blk), x.double() @ blk.double().t()) < 3e-6
    assert _rel(token_gemm.linear_bwd_data(dy, blk), dy.double() @ blk.double()) < 3e-6


def test_token_linear_autograd_uses_the_kernels_and_matches_torch():
    """The autograd op every MiT / head Linear goes through (segdistill_amd/linear.py) against torch's own Linear in fp64."""
    from segdistill_amd.linear import token_linear
    dev = torch.device('cuda:0')
    g = torch.Generator(device=dev).manual_seed(5)
    x = torch.randn(2, 8192, 64, device=dev, generator=g, requires_grad=True)
    w = (torch.randn(256, 64, device=dev, generator=g) / 8).requires_grad_(True)
    b = torch.randn(256, device=dev, generator=g).requires_grad_(True)
    up = torch.randn(2, 8192, 256, device=dev, generator=g)
    y = token_linear(x, w, b)
    y.backward(up)
    x64, w64, b64 = (t.detach().double().requires_grad_(True) for t in (x, w, b))
    y64 = torch.nn.functional.linear(x64, w64, b64)
    y64.backward(up.double())
    assert _rel(y, y64.detach()) < 3e-6
    assert _rel(x.grad, x64.grad) < 3e-6 and _rel(w.grad, w64.grad) < 3e-5 and _rel(b.grad, b64.grad) < 3e-5
    with torch.no_grad():     # the frozen teacher's path
        assert _rel(token_linear(x, w.detach(), b.detach()), y64.detach()) < 3e-6


# (.., 256 / 768, 19): <= 32 classes is ONE k-step of the input-gradient product = a single LDS staging buffer, smaller than the row-major
# epilogue image of wave 3 (round-2 advisor finding: rows 28-31 of its blocks were dropped when in_features > 64)
@pytest.mark.parametrize('case', [(2, 1024, 256, 150), (1, 4096, 64, 19), (3, 260, 32, 150), (8, 16384, 256, 150), (2, 1024, 256, 19), (1, 512, 768, 19),
                                  (2, 384, 128, 32)])
@pytest.mark.parametrize('split', [1, 0])
def test_linear_to_planes_matches_conv1x1(case, split):
    """sd_linear_nchw_* (the head's linear_pred on a token-major map, logits written as class planes, gradient read from planes) against
    the 1x1 conv it replaces in fp64: forward, input gradient, weight and bias gradients; both arithmetic modes."""
    from segdistill_amd import _lib, linear
    B, P, K, N = case
    g = torch.Generator().manual_seed(P + K + N)
    x = torch.randn(B, P, K, generator=g)
    w = torch.randn(N, K, generator=g) * K ** -0.5
    b = torch.randn(N, generator=g)
    dy = torch.randn(B, N, P, generator=g)
    x64, w64, b64 = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    ref = torch.einsum('bpk,nk->bnp', x64, w64) + b64[None, :, None]
    ref.backward(dy.double())
    dev = torch.device('cuda:0')
    _lib.set_tunable('align_split_bf16', split)
    try:
        xd, wd, bd = x.to(dev).requires_grad_(True), w.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
        assert linear.linear_to_planes_supported(xd, wd, bd)
        y = linear.linear_to_planes(xd, wd, bd)
        y.backward(dy.to(dev))
    finally:
        _lib.set_tunable('align_split_bf16', 1)
    def err(a, r):
        return float((a.double().cpu() - r).abs().max() / r.abs().max())
    assert y.shape == (B, N, P) and y.is_contiguous()
    assert err(y, ref.detach()) < 2e-6
    assert err(xd.grad, x64.grad) < 2e-6
    assert err(wd.grad, w64.grad) < 5e-6
    assert err(bd.grad, b64.grad) < 5e-6


@pytest.mark.parametrize('case', [(2, 1024, 256, 150), (1, 4096, 64, 19), (2, 16384, 768, 150)])
def test_linear_to_planes_bf16_storage(case):
    """bf16 activations, fp32 master weight / bias (config 5): against fp64 on the bf16-rounded operands (the kernel rounds the master weight
    to bf16 on its way into LDS, as autocast's cast does); outputs and input gradient are stored in bf16."""
    from segdistill_amd import linear
    B, P, K, N = case
    g = torch.Generator().manual_seed(P + K + N)
    x = torch.randn(B, P, K, generator=g).to(torch.bfloat16)
    w = torch.randn(N, K, generator=g) * K ** -0.5
    b = torch.randn(N, generator=g)
    dy = torch.randn(B, N, P, generator=g).to(torch.bfloat16)
    x64, w64, b64 = x.double().requires_grad_(True), w.to(torch.bfloat16).double().requires_grad_(True), b.double().requires_grad_(True)
    ref = torch.einsum('bpk,nk->bnp', x64, w64) + b64[None, :, None]
    ref.backward(dy.double())
    dev = torch.device('cuda:0')
    xd, wd, bd = x.to(dev).requires_grad_(True), w.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
    with torch.autocast('cuda', dtype=torch.bfloat16):
        assert linear.linear_to_planes_supported(xd, wd, bd)
        y = linear.linear_to_planes(xd, wd, bd)
    y.backward(dy.to(dev))
    def err(a, r):
        return float((a.detach().double().cpu() - r).abs().max() / r.abs().max())
    assert y.dtype == torch.bfloat16 and y.shape == (B, N, P) and y.is_contiguous()
    assert err(y, ref.detach()) < 1e-2
    assert xd.grad.dtype == torch.bfloat16 and err(xd.grad, x64.grad) < 1e-2
    assert wd.grad.dtype == torch.float32 and err(wd.grad, w64.grad) < 2e-3
    assert err(bd.grad, b64.grad) < 1e-4


# ---- round 3: split-bf16 products on a PRE-SPLIT weight (planes in the order the matrix cores consume them) ----------------------------------
PLANE_SHAPES = [  # tokens, in_features (K), out_features (N); K % 32 == 0 (forward), N % 32 == 0 for the input gradient
    (4096, 32, 128), (1000, 64, 256), (4099, 256, 256), (777, 160, 640), (512, 640, 160), (300, 256, 150), (2048, 512, 2048), (16384, 64, 256),
    (131, 1280, 320), (128, 32, 32), (5000, 128, 96),
]


@pytest.fixture(params=[128, 64])
def planes_tile(request):
    """Both row-tile heights of the planes GEMM (the launcher's own rule picks 64 rows below 384 tiles of 128 x 128, i.e. for every shape a test
    can afford): forced through the tunable, restored afterwards."""
    from segdistill_amd import _lib
    _lib.set_tunable('planes_tile', request.param)
    yield request.param
    _lib.set_tunable('planes_tile', 0)


@pytest.mark.parametrize('T,K,N', PLANE_SHAPES)
def test_planes_gemm_matches_fp64_at_the_split_mode_bound(T, K, N, planes_tile):
    """sd_presplit_multi + sd_linear_fwd_planes / sd_linear_bwd_data_planes: the same six bf16 products as mode 1 with the weight's planes
    written beforehand -- the SAME error bound as the exact-f32 path, error within 2x of it; ragged token counts, column counts that are not
    a multiple of the 32-column block (150, 160) or of the 128-column tile, bias, residual, a single k-step (K = 32); 128- and 64-row tiles."""
    from segdistill_amd import planes, token_gemm
    dev = torch.device('cuda:0')
    g = torch.Generator(device=dev).manual_seed(T * 5 + K * 3 + N)
    x = torch.randn(T, K, device=dev, generator=g)
    w = torch.randn(N, K, device=dev, generator=g) / K ** 0.5
    b = torch.randn(N, device=dev, generator=g)
    r = torch.randn(T, N, device=dev, generator=g)
    dy = torch.randn(T, N, device=dev, generator=g)
    x64, w64, b64 = x.double(), w.double(), b.double()
    ref = x64 @ w64.t() + b64
    tol = 3e-7 * max(8.0, K ** 0.5)
    pf = planes.get(w, 'fwd')
    y = token_gemm.linear_fwd_planes(x, w, pf, b)
    assert _rel(y, ref) < tol
    assert _rel(token_gemm.linear_fwd_planes(x, w, pf), x64 @ w64.t()) < tol
    assert _rel(token_gemm.linear_fwd_planes(x, w, pf, b, residual=r), ref + r.double()) < tol
    e3, e1 = float((y.double() - ref).norm() / ref.norm()), float((token_gemm.linear_fwd(x, w, b).double() - ref).norm() / ref.norm())
    assert e3 < 2.0 * e1 + 1e-8, (e3, e1)
    if N % 32 == 0:
        pb = planes.get(w, 'bwd')
        assert _rel(token_gemm.linear_bwd_data_planes(dy, w, pb), dy.double() @ w64) < 3e-7 * max(8.0, N ** 0.5)


def test_planes_hold_the_exact_three_way_split_in_fragment_order():
    """The planes buffer, decoded on the host: hi + mid + lo == w exactly (fp32 values whose 24 significand bits fit three bf16 terms) and
    element (block nb, k-step ks, plane, lane l, e) is W[32 nb + (l & 31)][16 ks + 8 (l >> 5) + e]; columns beyond the matrix are zero."""
    from segdistill_amd import planes
    dev = torch.device('cuda:0')
    g = torch.Generator(device=dev).manual_seed(11)
    N, K = 150, 96
    w = torch.randn(N, K, device=dev, generator=g)
    buf = planes.get(w, 'fwd').view(torch.bfloat16).reshape((N + 31) // 32, K // 16, 3, 64, 8).float().cpu()
    wc = w.cpu()
    lane = torch.arange(64)
    for nb in range(buf.shape[0]):
        for ks in range(buf.shape[1]):
            n = 32 * nb + (lane & 31)
            k = 16 * ks + 8 * (lane >> 5)
            want = torch.zeros(64, 8)
            ok = n < N
            want[ok] = torch.stack([wc[n[ok], k[ok] + e] for e in range(8)], 1)
            hi, mid, lo = buf[nb, ks, 0], buf[nb, ks, 1], buf[nb, ks, 2]
            assert torch.equal(hi, want.to(torch.bfloat16).float())
            assert torch.equal((hi.double() + mid.double() + lo.double()).float(), want)
    # bwd planes: B(k, n) = W[k][n]
    wb = torch.randn(64, 40, device=dev, generator=g)
    bb = planes.get(wb, 'bwd').view(torch.bfloat16).reshape(2, 4, 3, 64, 8).float().cpu()
    for nb in range(2):
        for ks in range(4):
            n, k = 32 * nb + (lane & 31), 16 * ks + 8 * (lane >> 5)
            want = torch.zeros(64, 8)
            ok = n < 40
            want[ok] = torch.stack([wb.cpu()[k[ok] + e, n[ok]] for e in range(8)], 1)
            assert torch.equal((bb[nb, ks, 0].double() + bb[nb, ks, 1].double() + bb[nb, ks, 2].double()).float(), want)


def test_planes_follow_the_weight_through_optimizer_steps_checkpoint_loads_and_views():
    """planes.py's one freshness rule: (a) HipAdamW writes parameters through raw pointers and rewrites the planes IN PLACE in the same step
    (same buffer address, new contents); (b) a torch-side write (copy_) bumps the version and the next use recomputes; (c) the column blocks
    of one wide parameter (linear_fuse) are separate entries of the same base and are all refreshed; (d) planes.sync rewrites at once."""
    from segdistill_amd import planes, token_gemm
    from segdistill_amd.engine.optim import HipAdamW
    from segdistill_amd.linear import token_linear
    dev = torch.device('cuda:0')
    g = torch.Generator(device=dev).manual_seed(2)
    wide = torch.nn.Parameter(torch.randn(128, 256, device=dev, generator=g) / 16)
    w2 = torch.nn.Parameter(torch.randn(256, 64, device=dev, generator=g) / 8)
    opt = HipAdamW([wide, w2], lr=1e-2, weight_decay=0.01)
    x = torch.randn(40000, 64, device=dev, generator=g)        # 313 x 2 tiles of 128 x 128: the split-bf16 dispatch of linear.py
    x2 = torch.randn(40000, 128, device=dev, generator=g)

    def run():
        y = token_linear(x, w2)                                 # [T, 256]
        a, b = wide[:, :128], wide[:, 128:]
        z = token_linear(x2, a) + token_linear(x2, b)
        return y, z

    def check(y, z):
        assert _rel(y, x.double() @ w2.detach().double().t()) < 3e-6
        assert _rel(z, x2.double() @ (wide.detach()[:, :128] + wide.detach()[:, 128:]).double().t()) < 3e-6

    y, z = run()
    check(y, z)
    ents = [e for e in planes._ENTRIES.values() if e.base() is wide or e.base() is w2]
    assert len(ents) == 3 and all(e.fwd is not None for e in ents)
    ptrs = [e.fwd.data_ptr() for e in ents]
    before = [e.fwd.clone() for e in ents]
    (y.sum() + z.sum()).backward()
    opt.step()                                                  # raw-pointer update + in-place refresh
    assert [e.fwd.data_ptr() for e in ents] == ptrs
    assert all(not torch.equal(e.fwd, b0) for e, b0 in zip(ents, before))
    with torch.no_grad():
        check(*run())
    with torch.no_grad():
        w2.copy_(torch.randn(256, 64, device=dev, generator=g) / 8)       # a checkpoint load: version bump -> recomputed at next use
        check(*run())
        wide.data.mul_(0.5)                                      # behind torch's back: no version bump ...
        assert planes.sync([wide]) == 2                          # ... sync rewrites its entries now
        check(*run())
    assert [e.fwd.data_ptr() for e in ents] == ptrs


@pytest.mark.parametrize('T,M,N', [(2048, 512, 256), (2048, 256, 512), (8192, 640, 160), (2048, 64, 1024), (2048, 64, 32), (4000, 160, 320),
                                   (1000, 20, 36), (96, 256, 256), (8192, 160, 640), (8192, 1024, 256)])
def test_splitk_weight_gradient_matches_fp64(T, M, N):
    """sd_linear_wgrad_splitk (round 3): dW = dY^T . X for few tokens / large weights, split-K slabs + the batched combine; both arithmetic
    modes (split-bf16 when tokens % 32 == 0, exact f32 MFMA otherwise); error bound of a T-term fp32 dot product."""
    import ctypes as C
    from segdistill_amd import _lib, deferred
    from segdistill_amd.ops import _stream_ptr
    L = _lib.lib()
    dev = torch.device('cuda:0')
    g = torch.Generator(device=dev).manual_seed(T + M + N)
    dy = torch.randn(T, M, device=dev, generator=g)
    x = torch.randn(T, N, device=dev, generator=g)
    ref = dy.double().t() @ x.double()
    ns = L.sd_linear_wgrad_splitk_slabs(T, M, N)
    assert ns >= 1
    for split in (1, 0):
        _lib.set_tunable('align_split_bf16', split)
        try:
            ws = torch.full((ns, M * N), float('nan'), device=dev)
            _lib.check(L.sd_linear_wgrad_splitk(dy.data_ptr(), x.data_ptr(), ws.data_ptr(), ws.numel() * 4, T, M, N, _stream_ptr()), 'splitk')
        finally:
            _lib.set_tunable('align_split_bf16', 1)
        out = torch.empty(M * N, device=dev)
        deferred.reduce_now(ws, out, M * N, ns)
        assert _rel(out.view(M, N), ref) < 3e-7 * max(8.0, T ** 0.5), (split, ns)
    assert L.sd_linear_wgrad_splitk_slabs(T, M + 1, N) == 0 and L.sd_linear_wgrad_splitk(dy.data_ptr(), x.data_ptr(), ws.data_ptr(), 16, T, M, N, None) != 0
    assert L.sd_linear_wgrad_splitk_slabs(2048, 1024, 256) == 0        # few tokens and 16 tiles: the library's turf (measured)


def test_token_linear_few_tokens_takes_the_splitk_weight_gradient():
    """The autograd op at a stage-4 shape (2048 tokens, 256 -> 512 with bias): the weight gradient goes through sd_linear_wgrad_splitk --
    immediately outside a deferred scope, through the batched combine inside one -- and matches fp64."""
    from segdistill_amd import deferred
    from segdistill_amd.linear import token_linear
    dev = torch.device('cuda:0')
    g = torch.Generator(device=dev).manual_seed(9)
    x = torch.randn(8, 256, 256, device=dev, generator=g, requires_grad=True)
    w = (torch.randn(512, 256, device=dev, generator=g) / 16).requires_grad_(True)
    b = torch.randn(512, device=dev, generator=g).requires_grad_(True)
    up = torch.randn(8, 256, 512, device=dev, generator=g)
    x64, w64, b64 = (t.detach().double().requires_grad_(True) for t in (x, w, b))
    torch.nn.functional.linear(x64, w64, b64).backward(up.double())
    for scoped in (False, True):
        x.grad = w.grad = b.grad = None
        y = token_linear(x, w, b, defer_ok=True)
        if scoped:
            with deferred.scope():
                y.backward(up)
        else:
            y.backward(up)
        assert _rel(w.grad, w64.grad) < 3e-5 and _rel(b.grad, b64.grad) < 3e-5 and _rel(x.grad, x64.grad) < 3e-6, scoped


@pytest.mark.parametrize('case', [(2, 1024, 256, 150), (1, 4096, 64, 19), (8, 16384, 768, 150), (3, 260, 32, 150), (2, 512, 96, 160), (1, 384, 128, 33)])
def test_linear_to_planes_forward_on_row_planes(case):
    """sd_linear_nchw_fwd_planes (round 3): W as pre-split row-major planes staged through LDS; against fp64 and against the round-2 forward
    (same six products per term: both within the split-mode bound), class counts that are not a multiple of 32, ragged pixel counts."""
    from segdistill_amd import _lib, planes
    from segdistill_amd.ops import _stream_ptr
    B, P, K, N = case
    dev = torch.device('cuda:0')
    g = torch.Generator(device=dev).manual_seed(P + K + N)
    x = torch.randn(B, P, K, device=dev, generator=g)
    w = torch.randn(N, K, device=dev, generator=g) * K ** -0.5
    b = torch.randn(N, device=dev, generator=g)
    ref = torch.einsum('bpk,nk->bnp', x.double(), w.double()) + b.double()[None, :, None]
    L = _lib.lib()
    y = torch.full((B, N, P), float('nan'), device=dev)
    pr = planes.get(w, 'rows')
    _lib.check(L.sd_linear_nchw_fwd_planes(x.data_ptr(), pr.data_ptr(), b.data_ptr(), y.data_ptr(), 0, B, P, K, N, _stream_ptr()), 'fwd_planes')
    y0 = torch.empty_like(y)
    _lib.check(L.sd_linear_nchw_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), y0.data_ptr(), 0, B, P, K, N, _stream_ptr()), 'fwd')
    tol = 3e-7 * max(8.0, K ** 0.5)
    assert _rel(y, ref) < tol and _rel(y0, ref) < tol
    # the row planes themselves: hi + mid + lo == w exactly, plane-major
    pl = pr.view(torch.bfloat16).reshape(3, N, K).double()
    assert torch.equal((pl[0] + pl[1] + pl[2]).float(), w) and torch.equal(pl[0].float(), w.to(torch.bfloat16).float())
