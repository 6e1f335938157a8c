"""MFMA 1x1 projection kernels vs a plain PyTorch reference of the same op (fp64 conv2d on CPU).
Tolerance: fp32 MFMA is an exact fmaf chain -> 2e-5 relative (max-norm scaled); bf16 storage -> 1e-2."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CASES = [
    # B, Cs, Ct, h, w
    (2, 128, 512, 16, 16),     # config-4 channel counts
    (1, 256, 768, 8, 8),       # config-5 channel counts (768 = 6 x 128)
    (2, 20, 37, 5, 7),         # nothing a multiple of the tile
    (1, 130, 129, 12, 11),     # tile edges + K tail
    (3, 16, 128, 32, 32),
]


def _ref(x, w, b, dy):
    x64 = x.double().requires_grad_(True)
    w64 = w.double().requires_grad_(True)
    b64 = b.double().requires_grad_(True)
    y = F.conv2d(x64, w64[:, :, None, None], b64)
    y.backward(dy.double())
    return y.detach(), x64.grad, w64.grad, b64.grad


def _err(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize('case', CASES)
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_align1x1_fwd_bwd(case, dtype):
    from segdistill_amd.align import align1x1
    B, Cs, Ct, h, w = case
    g = torch.Generator().manual_seed(B * 1000 + Cs)
    x = torch.randn(B, Cs, h, w, generator=g).to(dtype)
    wt = torch.randn(Ct, Cs, generator=g) / Cs ** 0.5
    b = torch.randn(Ct, generator=g)
    dy = torch.randn(B, Ct, h, w, generator=g).to(dtype)
    y_ref, dx_ref, dw_ref, db_ref = _ref(x.float(), wt, b, dy.float())
    dev = torch.device('cuda:0')
    xg = x.to(dev).requires_grad_(True)
    wg = wt.to(dev).requires_grad_(True)
    bg = b.to(dev).requires_grad_(True)
    y = align1x1(xg, wg, bg)
    y.backward(dy.to(dev))
    tol = 2e-5 if dtype == torch.float32 else 1e-2
    assert y.dtype == dtype and xg.grad.dtype == dtype and wg.grad.dtype == torch.float32
    assert _err(y, y_ref) < tol
    assert _err(xg.grad, dx_ref) < tol
    assert _err(wg.grad, dw_ref) < (2e-5 if dtype == torch.float32 else 2e-5)  # fp32 accumulation of the SAME rounded inputs
    assert _err(bg.grad, db_ref) < 2e-5


@pytest.mark.parametrize('case', [(8, 128, 512, 64, 64),     # BASELINE config 4: [8,128,64,64] -> 512 channels
                                  (2, 128, 512, 10, 10),     # a ragged last pixel tile (P = 100 = 64 + 36), one tile per workgroup
                                  (3, 48, 200, 12, 20),      # K = 48 (three k-steps), channel count not a multiple of 32 / 128
                                  (1, 16, 32, 2, 2),         # one pixel group
                                  (2, 96, 160, 36, 36)])     # P = 1296 = 20 tiles + 16 pixels
def test_streaming_forward_matches_fp64_and_the_generic_kernel(case):
    """csrc/align_stream.hip (round 6): the forward with W resident in registers, through the C ABI -- against fp64 (bound of the exact-f32 path)
    and against the generic pipelined GEMM it replaces (tunable align_stream = 0): same split-bf16 arithmetic, different summation order."""
    from segdistill_amd import _lib
    L = _lib.lib()
    B, Cs, Ct, h, w = case
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(Cs * 7 + Ct)
    x = torch.randn(B, Cs, h, w, generator=g).to(dev)
    wt = (torch.randn(Ct, Cs, generator=g) / Cs ** 0.5).to(dev)
    b = torch.randn(Ct, generator=g).to(dev)
    st = torch.cuda.current_stream().cuda_stream
    outs = []
    assert L.sd_get_tunable(b'align_stream') == 1
    try:
        for flag in (1, 0):
            assert L.sd_set_tunable(b'align_stream', flag) == 0
            y = torch.full((B, Ct, h, w), float('nan'), device=dev)
            assert L.sd_align1x1_fwd(x.data_ptr(), wt.data_ptr(), b.data_ptr(), y.data_ptr(), 0, B, Cs, Ct, h, w, st) == 0
            torch.cuda.synchronize()
            outs.append(y)
    finally:
        L.sd_set_tunable(b'align_stream', 1)
    ref = F.conv2d(x.double(), wt.double()[:, :, None, None], b.double())
    assert not torch.isnan(outs[0]).any()
    assert _err(outs[0], ref) < 2e-5 and _err(outs[1], ref) < 2e-5
    assert _err(outs[0], outs[1].double()) < 1e-5
    # no bias: the accumulators start from zero
    y = torch.empty(B, Ct, h, w, device=dev)
    assert L.sd_align1x1_fwd(x.data_ptr(), wt.data_ptr(), None, y.data_ptr(), 0, B, Cs, Ct, h, w, st) == 0
    assert _err(y, F.conv2d(x.double(), wt.double()[:, :, None, None])) < 2e-5


def test_align1x1_asymmetric_identity():
    """A = I with an asymmetric B catches a transposed C write (cdna guide, MFMA section)."""
    from segdistill_amd.align import align1x1
    dev = torch.device('cuda:0')
    C = 64
    x = torch.arange(C * 6, dtype=torch.float32).reshape(1, C, 2, 3).to(dev)
    y = align1x1(x, torch.eye(C, device=dev), None)
    assert torch.equal(y, x)
    perm = torch.randperm(C)
    y = align1x1(x, torch.eye(C, device=dev)[perm], None)
    assert torch.equal(y, x[:, perm.to(dev)])


def test_feature_align_module_in_distillation_loss():
    """channel_nums=(Cs,Ct) builds a trainable FeatureAlign whose grads flow through the HIP criterion."""
    import segdistill_amd
    from segdistill_amd.distillation import DistillationLoss
    segdistill_amd.register_all()
    dev = torch.device('cuda:0')
    dl = DistillationLoss([dict(student_layer='a', teacher_layer='b', loss_name='KLDLoss', channel_nums=(8, 16),
                                loss_config=dict(alpha=2, tau=2, transform_config={'loss_type': 'channel', 'group_size': 4}))]).to(dev)
    s = torch.randn(2, 8, 16, 16, device=dev, requires_grad=True)
    t = torch.randn(2, 16, 16, 16, device=dev)
    out = dl({'a': s}, {'b': t}, torch.zeros(2, 1, 16, 16, dtype=torch.long, device=dev), 1)
    (k, v), = out.items()
    v.backward()
    al = dl.aligns['0']
    assert al.weight.grad is not None and al.weight.grad.abs().sum() > 0 and s.grad.abs().sum() > 0
    # reference math on the CPU in fp64
    from oracle import kd_ref
    w64, b64 = al.weight.detach().double().cpu(), al.bias.detach().double().cpu()
    s64 = s.detach().double().cpu().requires_grad_(True)
    proj = F.conv2d(s64, w64[:, :, None, None], b64)
    ref = kd_ref.eager_kld(proj, t.double().cpu(), alpha=2, tau=2, loss_type='channel', group_size=4)
    ref.backward()
    assert float(v) == pytest.approx(float(ref), rel=2e-5)
    assert _err(s.grad, s64.grad) < 1e-4


@pytest.mark.parametrize('shape', [(131072, 32, 32), (32768, 128, 32), (5000, 37, 21), (8192, 256, 64), (4096, 512, 2048), (300, 16, 8),
                                   (20001, 100, 70), (16384, 256, 256), (9000, 33, 129), (8193, 64, 64),
                                   # bf16: csrc/wgrad_tn.hip (transposed LDS reads) -- ragged tiles, ragged k-splits, the config-5 align shape
                                   (5000, 40, 24), (777, 136, 72), (2048, 1024, 256), (65536, 768, 256), (8192, 1280, 320), (33, 8, 8),
                                   # ... and the 256 x 256 tiles of the ring kernel with both extents ragged
                                   (4096, 296, 264), (1024, 520, 256)])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_linear_wgrad_kernel(shape, dtype):
    """dW = dY^T . X (split-K MFMA) vs fp64 matmul of the same (rounded) operands."""
    from segdistill_amd import _lib
    from segdistill_amd.ops import _DT
    T, M, N = shape
    g = torch.Generator().manual_seed(T + M)
    dy = torch.randn(T, M, generator=g).to(dtype)
    x = torch.randn(T, N, generator=g).to(dtype)
    ref = dy.double().t() @ x.double()
    dev = torch.device('cuda:0')
    dyg, xg = dy.to(dev), x.to(dev)
    L = _lib.lib()
    dw = torch.empty(M, N, device=dev)
    fuse = bool(L.sd_linear_wgrad_fuses_bias_dtype(_DT[dtype], T, M, N))
    db = torch.empty(M, device=dev) if fuse else None
    wsb = L.sd_linear_wgrad_workspace_bytes(T, M, N)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    rc = L.sd_linear_wgrad(dyg.data_ptr(), xg.data_ptr(), dw.data_ptr(), None if db is None else db.data_ptr(), _DT[dtype], T, M, N,
                           ws.data_ptr(), wsb, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    assert _err(dw, ref) < 2e-5
    if fuse:
        assert _err(db, dy.double().sum(0)) < 2e-5
    else:  # the tiled kernel does not produce the bias gradient and says so
        assert L.sd_linear_wgrad(dyg.data_ptr(), xg.data_ptr(), dw.data_ptr(), dw.data_ptr(), _DT[dtype], T, M, N, ws.data_ptr(), wsb, None) == -6


@pytest.mark.parametrize('shape', [(16384, 256, 256), (4096, 296, 264), (1024, 520, 256), (2048, 1024, 256), (8192, 1280, 320), (65536, 768, 256),
                                   (131072, 768, 256)])
def test_linear_wgrad_tn_big_tiles(shape):
    """csrc/wgrad_tn.hip, 256 x 256 tiles of the LDS-DMA ring (product rule: only long k ranges; tunable 2 = wherever legal) vs fp64 and vs the
    128 x 128 tiles: ragged tile rows / columns, ragged last k-split, both ring depths of the prologue."""
    from segdistill_amd import _lib
    T, M, N = shape
    g = torch.Generator().manual_seed(T + M)
    dy = torch.randn(T, M, generator=g).to(torch.bfloat16)
    x = torch.randn(T, N, generator=g).to(torch.bfloat16)
    dev = torch.device('cuda:0')
    dyg, xg = dy.to(dev), x.to(dev)
    ref = dyg.double().t() @ xg.double()
    L = _lib.lib()
    old = _lib.get_tunable('wgrad_tn_ring')
    out = {}
    try:
        for mode in (1, 2):
            _lib.set_tunable('wgrad_tn_ring', mode)
            dw = torch.full((M, N), float('nan'), device=dev)
            wsb = L.sd_linear_wgrad_workspace_bytes(T, M, N)
            ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
            assert L.sd_linear_wgrad(dyg.data_ptr(), xg.data_ptr(), dw.data_ptr(), None, 1, T, M, N, ws.data_ptr(), wsb,
                                     torch.cuda.current_stream().cuda_stream) == 0
            out[mode] = dw
            assert _err(dw, ref) < 2e-5
    finally:
        _lib.set_tunable('wgrad_tn_ring', old)
    assert _err(out[2], out[1]) < 1e-6


def test_token_linear_autograd_matches_f_linear():
    from segdistill_amd.linear import token_linear
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    x = torch.randn(2, 4096, 48, device=dev, requires_grad=True)
    w = torch.randn(96, 48, device=dev, requires_grad=True)
    b = torch.randn(96, device=dev, requires_grad=True)
    y = token_linear(x, w, b)
    y.square().mean().backward()
    g = (x.grad.clone(), w.grad.clone(), b.grad.clone())
    x.grad = w.grad = b.grad = None
    torch.nn.functional.linear(x, w, b).square().mean().backward()
    for a, r in zip(g, (x.grad, w.grad, b.grad)):
        assert _err(a, r) < 1e-4


@pytest.mark.parametrize('shape', [(2048, 64, 4096), (2048, 32, 2048), (500, 37, 1030), (2048, 320, 1280)])
def test_sr_patch_linear(shape):
    """The SR patch projection (a Linear over the gathered r x r patches, reduction axis up to 4096) through token_linear vs fp64, forward and gradients."""
    from segdistill_amd.linear import sr_patch_linear
    M, N, K = shape
    g = torch.Generator().manual_seed(M + N)
    x = torch.randn(4, M // 4, K, generator=g)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    dev = torch.device('cuda:0')
    xg, wg, bg = x.to(dev).requires_grad_(True), w.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
    y = sr_patch_linear(xg, wg, bg)
    ref = torch.nn.functional.linear(x.double(), w.double(), b.double())
    assert _err(y, ref) < 2e-5
    y.square().mean().backward()
    x64, w64, b64 = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    torch.nn.functional.linear(x64, w64, b64).square().mean().backward()
    assert _err(xg.grad, x64.grad) < 1e-4 and _err(wg.grad, w64.grad) < 1e-4 and _err(bg.grad, b64.grad) < 1e-4


def test_token_linear_under_bf16_autocast_matches_f_linear():
    """Under autocast the split-K weight-gradient path computes what F.linear + autograd compute: bf16 forward, fp32 master
    gradients (checked against the fp64 product of the bf16-rounded operands)."""
    from segdistill_amd.linear import _TokenLinear, token_linear
    import torch.nn.functional as F
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    x = torch.randn(4, 4096, 64, device=dev, requires_grad=True)
    w = (torch.randn(96, 64, device=dev) * 0.1).requires_grad_(True)
    b = torch.randn(96, device=dev).requires_grad_(True)
    dy = torch.randn(4, 4096, 96, device=dev)
    calls, real = [], _TokenLinear.apply
    _TokenLinear.apply = lambda *a: (calls.append(1), real(*a))[1]
    try:
        with torch.autocast('cuda', dtype=torch.bfloat16):
            y = token_linear(x, w, b)
    finally:
        _TokenLinear.apply = real
    assert len(calls) == 1 and y.dtype == torch.bfloat16
    y.backward(dy.to(torch.bfloat16))
    got = (x.grad.clone(), w.grad.clone(), b.grad.clone())
    assert got[0].dtype == torch.float32 and got[1].dtype == torch.float32
    x.grad = w.grad = b.grad = None
    with torch.autocast('cuda', dtype=torch.bfloat16):
        y2 = F.linear(x, w, b)
    y2.backward(dy.to(torch.bfloat16))
    assert torch.equal(y, y2)
    xb, wb, gb = x.detach().bfloat16().double(), w.detach().bfloat16().double(), dy.bfloat16().double()
    dw64 = gb.reshape(-1, 96).t() @ xb.reshape(-1, 64)
    db64 = gb.reshape(-1, 96).sum(0)

    def rel(a, r):
        return float((a.double() - r).abs().max() / r.abs().max())
    assert rel(got[1], dw64) < 2e-5 and rel(got[2], db64) < 2e-5          # fp32 accumulation of bf16 products: tighter than the library
    assert rel(w.grad, dw64) < 2e-2 and rel(got[0], x.grad.double()) < 2e-2


@pytest.mark.parametrize('shape', [(131072, 256, 256), (65536, 256, 160), (16384, 256, 64), (8192, 256, 32), (9001, 136, 40), (8192, 128, 72),
                                   (32768, 512, 128), (8200, 384, 264)])
def test_linear_wgrad_tn_split_bf16(shape):
    """csrc/wgrad_tn.hip, fp32 storage: dW = dY^T . X as split-K slabs on transposed LDS reads in split-bf16 arithmetic against the fp64
    product -- held to the exact-f32 tall-skinny kernel's own error level (both accumulate ~T products in fp32): rel-L2 within 2x of it and
    under 4e-6 -- incl. ragged token counts, out_features that are not a multiple of the 128-row tile and all three tile widths."""
    from segdistill_amd import _lib, deferred
    T, M, N = shape
    g = torch.Generator().manual_seed(T + 7 * M + N)
    dy = torch.randn(T, M, generator=g)
    x = torch.randn(T, N, generator=g)
    ref = dy.double().t() @ x.double()
    dev = torch.device('cuda:0')
    dyg, xg = dy.to(dev), x.to(dev)
    L = _lib.lib()
    ns = L.sd_linear_wgrad_tn_slabs(T, M, N)
    assert ns > 0
    slabs = torch.empty(ns, M * N, device=dev)
    out = torch.empty(M * N, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    assert L.sd_linear_wgrad_tn(dyg.data_ptr(), xg.data_ptr(), slabs.data_ptr(), slabs.numel() * 4, T, M, N, 0, st) == 0
    deferred.reduce_now(slabs, out, M * N, ns)
    e_tn = _err(out.view(M, N), ref)
    # with the bias gradient riding along: M more floats per slab = fp32 column sums of dY; the weight part is the same bits
    slabs_b = torch.full((ns, M * N + M), float('nan'), device=dev)
    out_b = torch.empty(M * N + M, device=dev)
    assert L.sd_linear_wgrad_tn(dyg.data_ptr(), xg.data_ptr(), slabs_b.data_ptr(), slabs_b.numel() * 4, T, M, N, 1, st) == 0
    deferred.reduce_now(slabs_b, out_b, M * N + M, ns)
    assert torch.equal(out_b[:M * N], out)
    assert _err(out_b[M * N:], dy.double().sum(0)) < 2e-5
    assert L.sd_linear_wgrad_tn(dyg.data_ptr(), xg.data_ptr(), slabs.data_ptr(), slabs.numel() * 4, T, M, N, 1, st) == -4      # workspace without room for it
    dw = torch.empty(M, N, device=dev)
    wsb = L.sd_linear_wgrad_workspace_bytes(T, M, N)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    assert L.sd_linear_wgrad(dyg.data_ptr(), xg.data_ptr(), dw.data_ptr(), None, 0, T, M, N, ws.data_ptr(), wsb, st) in (0, -6)
    e_f32 = _err(dw, ref)
    assert e_tn < 2e-5 and e_tn < 2 * e_f32 + 5e-7, (e_tn, e_f32)          # 2e-5: the bar test_linear_wgrad_kernel holds the exact-f32 kernel to
    # not this kernel's shapes say so
    assert L.sd_linear_wgrad_tn_slabs(2048, 256, 256) == 0 and L.sd_linear_wgrad_tn_slabs(131072, 64, 64) == 0 and L.sd_linear_wgrad_tn_slabs(131072, 252, 64) == 0
