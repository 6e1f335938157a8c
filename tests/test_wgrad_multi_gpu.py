"""The grouped bf16 weight-gradient launch (csrc/wgrad_tn.hip::wgrad_tn_bf16_ring_multi; autograd's `grad_output.t().mm(input)` of every token-major
Linear, mix_transformer.py:24-27,48-55,75-84,107-133, segformer_head.py:22-33, under bf16 storage): all weight gradients of a deferred scope in one
launch with jointly planned k-splits, against fp64 on the same bf16 operands; one-split products written straight to their destination; more jobs
than one kernel table holds; run-to-run bit identity (fixed reduction order); the joint plan's slab bytes against the per-product plans'."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu

# tokens, out, in -- the B1 student's Linears of BASELINE config 5 (a block per stage, head projections, align)
SHAPES = [(131072, 256, 64), (131072, 64, 256), (2048, 128, 64), (32768, 128, 128), (32768, 512, 128), (2048, 256, 128), (8192, 320, 320),
          (8192, 1280, 320), (8192, 320, 1280), (2048, 640, 320), (2048, 512, 512), (2048, 2048, 512), (2048, 512, 2048), (2048, 1024, 512),
          (131072, 768, 256), (32768, 768, 256), (8192, 768, 256), (2048, 768, 256), (96, 8, 8), (4096, 72, 40)]


def _operands(dev, shapes, seed=0):
    g = torch.Generator(device='cpu').manual_seed(seed)
    out = []
    for T, M, N in shapes:
        dy = (torch.randn(T, M, generator=g) * 0.5).to(dev).bfloat16()
        x = torch.randn(T, N, generator=g).to(dev).bfloat16()
        out.append((dy, x))
    return out


def _grouped(ops, shapes):
    from segdistill_amd import deferred, linear
    res = []
    with deferred.scope():
        for (dy, x), (T, M, N) in zip(ops, shapes):
            assert deferred.wgrad_groupable(dy, x, M, N)
            dw, db = linear.linear_weight_grads(x, dy, (M, N), torch.float32, True, True, True)
            res.append((dw, db))
    torch.cuda.synchronize()
    return res


def test_grouped_launch_matches_fp64_and_is_run_to_run_identical():
    dev = torch.device('cuda:0')
    ops = _operands(dev, SHAPES)
    a = _grouped(ops, SHAPES)
    b = _grouped(ops, SHAPES)
    for (dy, x), (T, M, N), (dw, db), (dw2, db2) in zip(ops, SHAPES, a, b):
        ref = dy.double().t() @ x.double()
        scale = float(ref.abs().max()) + 1e-12
        assert dw.shape == (M, N) and dw.dtype == torch.float32
        assert float((dw.double() - ref).abs().max()) <= 2e-5 * scale * max(1.0, (T / 4096) ** 0.5), (T, M, N)
        refb = dy.double().sum(0)
        assert float((db.double() - refb).abs().max()) <= 1e-4 * (float(refb.abs().max()) + 1.0)
        assert torch.equal(dw, dw2) and torch.equal(db, db2)


def test_more_jobs_than_one_table_and_the_ungrouped_switch(monkeypatch):
    """40 jobs = two kernel tables; and SEGDISTILL_WGRAD_GROUPED=0's path (each product launched on its own inside the backward) gives the same
    values up to the k-split boundaries' summation order."""
    from segdistill_amd import deferred
    dev = torch.device('cuda:0')
    shapes = [(2048 + 1024 * (i % 3), 64 + 32 * (i % 5), 64 + 64 * (i % 4)) for i in range(40)]
    ops = _operands(dev, shapes, seed=5)
    a = _grouped(ops, shapes)
    monkeypatch.setattr(deferred, '_WGRAD_GROUPED', False)
    from segdistill_amd import linear
    with deferred.scope():
        b = [linear.linear_weight_grads(x, dy, (M, N), torch.float32, False, True, True)[0] for (dy, x), (T, M, N) in zip(ops, shapes)]
    torch.cuda.synchronize()
    for (dw, _), dw2, (dy, x) in zip(a, b, ops):
        ref = dy.double().t() @ x.double()
        tol = 3e-5 * (float(ref.abs().max()) + 1e-12)
        assert float((dw.double() - ref).abs().max()) <= tol and float((dw2.double() - ref).abs().max()) <= tol


def test_joint_plan_cuts_the_slab_bytes():
    """Host-side: the joint plan of config 5's bf16 weight gradients writes a fraction of the slab bytes of the per-product plans (VERDICT r4 item 4:
    832 MB next to 1478 MB of operands), products over 2048 tokens get a single split, and the whole launch stays near its workgroup target."""
    from segdistill_amd import _lib, deferred
    L = _lib.lib()
    arr = (deferred._WgradJob * len(SHAPES))()
    for k, (T, M, N) in enumerate(SHAPES):
        arr[k].tokens, arr[k].out_features, arr[k].in_features = T, M, N
    assert L.sd_linear_wgrad_tn_multi_plan(C.cast(arr, C.c_void_p), len(SHAPES), 1) == 0
    joint = sum(arr[k].nsplit * M * N * 4 for k, (T, M, N) in enumerate(SHAPES) if arr[k].nsplit > 1)
    single = sum(max(L.sd_linear_wgrad_generic_slabs(1, T, M, N), 1) * M * N * 4 for T, M, N in SHAPES)
    wgs = sum(arr[k].nsplit * -(-M // 128) * -(-N // 128) for k, (T, M, N) in enumerate(SHAPES))
    assert joint < 0.5 * single and 512 <= wgs <= 4096
    assert all(arr[k].nsplit >= 1 for k in range(len(SHAPES)))
    assert all(arr[k].nsplit == 1 for k, (T, M, N) in enumerate(SHAPES) if T <= 2048 and M * N >= 512 * 512)


# fp32 storage: the B0 student's Linears of BASELINE config 2 (tokens, out, in), incl. out < 128 (half-empty tile rows), the three tile widths,
# few-token stages, a ragged token count and a bias-free product
SHAPES_F32 = [(131072, 32, 32), (131072, 128, 32), (131072, 32, 128), (2048, 64, 32), (32768, 64, 64), (32768, 256, 64), (32768, 64, 256),
              (8192, 160, 160), (8192, 640, 160), (8192, 160, 640), (2048, 320, 160), (2048, 256, 256), (2048, 1024, 256), (2048, 256, 1024),
              (2048, 512, 256), (131072, 256, 32), (32768, 256, 64), (8192, 256, 160), (2048, 256, 256), (131072, 256, 256), (9001, 136, 40)]


def _rel_l2(a, r):
    return float(((a.double() - r) ** 2).sum().sqrt() / (r ** 2).sum().sqrt().clamp_min(1e-300))


def test_fp32_grouped_launches_match_fp64_with_the_bias_sums_riding_along():
    """wgrad_tn_x3_multi: one launch per tile width for all fp32 weight gradients of a scope, in split-bf16 arithmetic (fp32-grade: held to the bar of
    tests/test_align_gpu.py::test_linear_wgrad_tn_split_bf16), bias gradients as M extra floats per slab, run-to-run identical."""
    from segdistill_amd import deferred, linear
    dev = torch.device('cuda:0')
    g = torch.Generator(device='cpu').manual_seed(11)
    ops = [(torch.randn(T, M, generator=g).to(dev), torch.randn(T, N, generator=g).to(dev)) for T, M, N in SHAPES_F32]

    def run():
        res = []
        with deferred.scope():
            for k, ((dy, x), (T, M, N)) in enumerate(zip(ops, SHAPES_F32)):
                assert deferred.wgrad_groupable(dy, x, M, N)
                res.append(linear.linear_weight_grads(x, dy, (M, N), torch.float32, k % 5 != 4, True, True))
        torch.cuda.synchronize()
        return res
    a, b = run(), run()
    for k, ((dy, x), (T, M, N), (dw, db), (dw2, db2)) in enumerate(zip(ops, SHAPES_F32, a, b)):
        ref = dy.double().t() @ x.double()
        assert dw.shape == (M, N) and _rel_l2(dw, ref) < 2e-5, (T, M, N, _rel_l2(dw, ref))
        assert torch.equal(dw, dw2)
        if k % 5 != 4:
            assert db.shape == (M,) and _rel_l2(db, dy.double().sum(0)) < 2e-5 and torch.equal(db, db2)
        else:
            assert db is None


def test_fp32_student_step_is_the_same_grouped_and_one_by_one(monkeypatch):
    """A small fp32 MiT block stack, forward + backward inside a deferred scope: parameter gradients with the grouped launches (default) against
    the one-launch-per-Linear path of rounds 2-4 -- the same values up to the k-split boundaries' summation order."""
    from segdistill_amd import deferred, linear
    from segdistill_amd.backbones.mit import MixVisionTransformer
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    net = MixVisionTransformer(embed_dims=(32, 64, 160, 256), num_heads=(1, 2, 5, 8), depths=(1, 1, 1, 1), sr_ratios=(8, 4, 2, 1), drop_path_rate=0.0).to(dev)
    img = torch.randn(2, 3, 256, 256, device=dev)

    def grads():
        net.zero_grad(set_to_none=True)
        with deferred.scope():
            sum(f.square().mean() for f in net(img)).backward()
        torch.cuda.synchronize()
        return {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
    ga = grads()
    monkeypatch.setattr(linear, '_FP32_WGRAD_GROUPED', False)
    gb = grads()
    assert ga.keys() == gb.keys() and len(ga) > 40
    # rel-L2 per tensor.  Two runs of the SAME path already differ in the last bits (MIOpen's patch-embed convolution kernels accumulate with atomics),
    # so: 1e-3, plus an absolute floor for tensors whose gradient is noise-level next to the others (a bias in front of a normalisation)
    top = max(float(g.norm()) for g in gb.values())
    for n in ga:
        assert float((ga[n] - gb[n]).norm()) <= 1e-3 * float(gb[n].norm()) + 1e-7 * top, n

@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_depthwise_filter_gradients_grouped_equal_the_single_launches(dtype):
    """csrc/dwconv.hip::dw3x3_wgrad_partials_multi: the depth-wise 3 x 3 filter / bias gradients of a scope (Mix-FFN, mix_transformer.py:376-387) in
    ONE launch -- the same workgroup program per job as the single launch, hence bit-identical gradients; shapes of all four MiT stages + a ragged one."""
    from segdistill_amd import deferred, dwconv
    dev = torch.device('cuda:0')
    g = torch.Generator(device='cpu').manual_seed(21)
    cases = [(2, 64, 64, 128), (2, 32, 32, 256), (2, 16, 16, 640), (2, 8, 8, 1024), (1, 5, 7, 40), (3, 16, 16, 64)]
    mods = []
    for B, H, W, C in cases:
        w = torch.randn(C, 1, 3, 3, generator=g).to(dev).requires_grad_(True)
        b = torch.randn(C, generator=g).to(dev).requires_grad_(True)
        x = torch.randn(B, H * W, C, generator=g).to(dev).to(dtype)
        gy = torch.randn(B, H * W, C, generator=g).to(dev).to(dtype)
        mods.append((w, b, x, gy, H, W))

    def run(grouped):
        for w, b, *_ in mods:
            w.grad = b.grad = None
        deferred._WGRAD_GROUPED = grouped           # False: every partials launch on its own inside the backward (same combine at the scope's end)
        with deferred.scope():
            for w, b, x, gy, H, W in mods:
                with torch.autocast('cuda', dtype=torch.bfloat16, enabled=dtype == torch.bfloat16):
                    y = dwconv.dwconv3x3_tokens(x, w, b, H, W)
                y.backward(gy.to(y.dtype))
        torch.cuda.synchronize()
        return [(w.grad.clone(), b.grad.clone()) for w, b, *_ in mods]
    try:
        a, c = run(True), run(False)
    finally:
        deferred._WGRAD_GROUPED = True
    for (wa, ba), (wc, bc), (w, b, x, gy, H, W) in zip(a, c, mods):
        assert torch.equal(wa, wc) and torch.equal(ba, bc)
        B, C = x.shape[0], x.shape[2]
        ref = torch.nn.grad.conv2d_weight(x.double().transpose(1, 2).reshape(B, C, H, W), (C, 1, 3, 3), gy.double().transpose(1, 2).reshape(B, C, H, W),
                                          padding=1, groups=C)
        assert float((wa.double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max()) * (1 if dtype == torch.float32 else 4)


def test_held_operand_budget_flushes_early_and_changes_no_bit(monkeypatch):
    """the operands a scope keeps alive for its grouped launches are bounded (deferred.held_budget_bytes); past the budget the launches
    registered so far run at once.  A zero budget (every registration flushes = rounds 2-4's memory profile) must give the very same bits for the
    one-split products and fp64-grade values for all, and the bookkeeping must return to zero when the scope ends."""
    from segdistill_amd import deferred
    dev = torch.device('cuda:0')
    shapes = SHAPES[:10]
    ops = _operands(dev, shapes, seed=3)
    before = deferred.partial_flushes
    a = _grouped(ops, shapes)
    assert deferred.partial_flushes == before and deferred.held_bytes() == 0          # the default budget is never reached here
    monkeypatch.setattr(deferred, '_HELD_BUDGET_MB', '64')
    b = _grouped(ops, shapes)
    assert deferred.partial_flushes > before and deferred.held_bytes() == 0
    for (dy, x), (T, M, N), (dw, db), (dw2, db2) in zip(ops, shapes, a, b):
        ref = dy.double().t() @ x.double()
        scale = float(ref.abs().max()) + 1e-12
        assert float((dw2.double() - ref).abs().max()) <= 2e-5 * scale * max(1.0, (T / 4096) ** 0.5), (T, M, N)
        assert float((dw2.double() - dw.double()).abs().max()) <= 2e-5 * scale * max(1.0, (T / 4096) ** 0.5)
        assert torch.allclose(db, db2, rtol=1e-5, atol=1e-4)
