"""engine/optim.py::HipAdamW (csrc/optim.hip: one launch over every tensor) against torch.optim.AdamW's single-tensor reference path
(torch/optim/adamw.py::_single_tensor_adamw -- what the reference's mmcv OptimizerHook runs): parameters and both moments after several steps
with a changing learning rate, two parameter groups (lr x10 / no decay), a channels-last filter, ragged sizes, a tensor that gets no gradient,
and a state_dict round trip into a plain torch.optim.AdamW."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def _params(dev, seed):
    g = torch.Generator().manual_seed(seed)
    shapes = [(4097,), (33, 7), (256, 64), (5,), (1,), (64, 32, 3, 3), (300, 257)]
    ps = []
    for i, s in enumerate(shapes):
        p = torch.randn(*s, generator=g).to(dev)
        if len(s) == 4:
            p = p.contiguous(memory_format=torch.channels_last)
        ps.append(torch.nn.Parameter(p))
    return ps


def _grads(ps, step, skip_last, skip_first=False):
    g = torch.Generator().manual_seed(1000 + step)
    for i, p in enumerate(ps):
        if (skip_last and i == len(ps) - 1) or (skip_first and i == 0):
            p.grad = None
            torch.randn(*p.shape, generator=g)                    # keep the stream of the other tensors' gradients unchanged
            continue
        gr = torch.randn(*p.shape, generator=g).to(p.device)
        if p.dim() == 4:
            gr = gr.contiguous(memory_format=torch.channels_last)
        p.grad = gr


def test_one_launch_adamw_matches_torch():
    from segdistill_amd.engine.optim import HipAdamW
    dev = torch.device('cuda:0')
    a, b = _params(dev, 3), _params(dev, 3)
    groups = lambda ps: [dict(params=ps[:4], lr=6e-5, weight_decay=0.01), dict(params=ps[4:], lr=6e-4, weight_decay=0.0)]
    opt_a = HipAdamW(groups(a), lr=6e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    opt_b = torch.optim.AdamW(groups(b), lr=6e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, foreach=False, fused=False)
    for step in range(8):
        for opt in (opt_a, opt_b):
            for gi, g in enumerate(opt.param_groups):
                g['lr'] = (6e-5 if gi == 0 else 6e-4) * (1 - step / 10)          # the poly schedule moves it every iteration
        # the last tensor joins at step 2; the first one sits out steps 4 and 5 (two consecutive steps with an unchanged set of tensors)
        _grads(a, step, skip_last=step < 2, skip_first=step in (4, 5))
        _grads(b, step, skip_last=step < 2, skip_first=step in (4, 5))
        opt_a.step()
        opt_b.step()
    for pa, pb in zip(a, b):
        assert pa.stride() == pb.stride()
        assert torch.allclose(pa, pb, rtol=2e-6, atol=1e-7), float((pa - pb).abs().max())
        sa, sb = opt_a.state[pa], opt_b.state[pb]
        assert torch.allclose(sa['exp_avg'], sb['exp_avg'], rtol=2e-6, atol=1e-9)
        assert torch.allclose(sa['exp_avg_sq'], sb['exp_avg_sq'], rtol=2e-6, atol=1e-12)
    # the state dict is torch's: a plain AdamW resumes from it and both continue identically
    sd = copy.deepcopy(opt_a.state_dict())
    assert sorted(float(s['step']) for s in sd['state'].values()) == [6.0, 6.0, 8.0, 8.0, 8.0, 8.0, 8.0]   # torch counts per tensor
    c = [torch.nn.Parameter(p.detach().clone()) for p in a]
    opt_c = torch.optim.AdamW(groups(c), lr=6e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, foreach=False, fused=False)
    opt_c.load_state_dict(copy.deepcopy(sd))          # load_state_dict keeps the tensors it is handed when dtype and device already fit
    opt_d = HipAdamW(groups(a), lr=6e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    opt_d.load_state_dict(copy.deepcopy(sd))
    _grads(a, 9, False)
    _grads(c, 9, False)
    opt_d.step()
    opt_c.step()
    for pa, pc in zip(a, c):
        assert torch.allclose(pa, pc, rtol=2e-6, atol=1e-7)


def test_bf16_shadows_follow_the_parameters():
    """linear.lowp_copy keeps a bf16 shadow on a trainable parameter; HipAdamW rewrites it with every step (csrc/optim.hip), so a bf16-autocast
    forward never casts the weight again.  The shadow must equal the freshly cast parameter bit for bit, survive many steps, and be remade
    after a torch op writes the parameter."""
    from segdistill_amd.engine.optim import HipAdamW
    from segdistill_amd.linear import lowp_copy
    dev = torch.device('cuda:0')
    ps = _params(dev, 11)
    opt = HipAdamW([dict(params=ps, lr=1e-3, weight_decay=0.01)], lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    shadowed = ps[:3] + ps[5:6]                                    # incl. the channels-last filter
    shadows = [lowp_copy(p, torch.bfloat16) for p in shadowed]
    assert all(s.dtype == torch.bfloat16 and s.stride() == p.stride() for s, p in zip(shadows, shadowed))
    for step in range(4):
        _grads(ps, step, False)
        opt.step()
        for p, s in zip(shadowed, shadows):
            assert lowp_copy(p, torch.bfloat16) is s                # still the same tensor: nothing was cast again
            assert torch.equal(s, p.detach().to(torch.bfloat16))
    with torch.no_grad():
        shadowed[0].mul_(2.0)                                        # a torch op bumps the version: the shadow is stale and is remade
    s2 = lowp_copy(shadowed[0], torch.bfloat16)
    assert s2 is not shadows[0] and torch.equal(s2, shadowed[0].detach().to(torch.bfloat16))
    _grads(ps, 9, False)
    opt.step()                                                       # the optimizer follows the new shadow
    assert torch.equal(s2, shadowed[0].detach().to(torch.bfloat16))
