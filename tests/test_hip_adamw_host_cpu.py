"""Host logic of segdistill_amd.engine.optim.HipAdamW on the CPU: the descriptor table (one 56-byte record per tensor), the per-tensor
step counts ("missed" steps of tensors that sit a step out), per-group lr / weight decay and the state_dict hand-over -- driven through a
numpy EMULATION of sd_adamw_multi that reads the same table the HIP kernel reads (include/segdistill_hip.h) -- against torch.optim.AdamW.
The kernel itself is checked on the GPU (tests/test_hip_adamw_gpu.py)."""
import ctypes
import math

import numpy as np
import pytest
import torch

from segdistill_amd.engine import optim as sd_optim

CHUNK = 4096


class _FakeLib:
    """What step() needs from the C ABI; sd_adamw_multi applies torch's single-tensor AdamW formulas through the raw pointers of the table."""
    def __init__(self):
        self.calls = []

    def sd_adamw_chunk(self):
        return CHUNK

    def sd_adamw_max_groups(self):
        return 8

    def sd_adamw_multi(self, tensors, blocks, nblocks, lrs, ngroups, b1, b2, eps, step, stream):
        desc = np.frombuffer((ctypes.c_uint8 * (56 * self.ntensors)).from_address(tensors), dtype=sd_optim.HipAdamW._DESC)
        blk = np.frombuffer((ctypes.c_int32 * (2 * nblocks)).from_address(blocks), dtype=np.int32).reshape(-1, 2)
        assert sorted(set(blk[:, 0].tolist())) == list(range(self.ntensors))                 # every tensor has blocks ...
        for ti, d in enumerate(desc):
            assert (blk[:, 0] == ti).sum() == (int(d['n']) + CHUNK - 1) // CHUNK              # ... exactly ceil(n / chunk) of them
            n, gi, missed = int(d['n']), int(d['gm']) & 0xFF, int(d['gm']) >> 8
            arr = lambda ptr: np.frombuffer((ctypes.c_float * n).from_address(int(ptr)), dtype=np.float32)
            p, g, m, v = arr(d['p']), arr(d['g']), arr(d['m']), arr(d['v'])
            lr, wd, k = float(lrs[gi]), float(d['wd']), step - missed
            assert k >= 1
            p *= np.float32(1.0 - lr * wd)
            m += (g - m) * np.float32(1.0 - b1)
            v *= np.float32(b2)
            v += g * g * np.float32(1.0 - b2)
            bc1, bc2 = 1.0 - b1 ** k, 1.0 - b2 ** k
            denom = np.sqrt(v) / np.float32(math.sqrt(bc2)) + np.float32(eps)
            p -= np.float32(lr / bc1) * (m / denom)
        self.calls.append((step, self.ntensors, nblocks))
        return 0


@pytest.fixture
def fake(monkeypatch):
    from segdistill_amd import _lib, ops
    lib = _FakeLib()
    monkeypatch.setattr(_lib, 'lib', lambda: lib)
    monkeypatch.setattr(ops, '_stream_ptr', lambda: None)
    return lib


def _params(seed):
    g = torch.Generator().manual_seed(seed)
    shapes = [(5000,), (64, 3, 3, 3), (7,), (130, 33)]
    return [torch.nn.Parameter(torch.randn(*s, generator=g)) for s in shapes]


def _groups(ps):
    return [dict(params=ps[:2], lr=1e-2, weight_decay=0.01), dict(params=ps[2:], lr=3e-2, weight_decay=0.0)]


def test_steps_match_torch_adamw_with_tensors_sitting_steps_out(fake):
    a, b = _params(0), _params(0)
    hip = sd_optim.HipAdamW(_groups(a), betas=(0.9, 0.999), eps=1e-8)
    ref = torch.optim.AdamW(_groups(b), betas=(0.9, 0.999), eps=1e-8, foreach=False)
    gen = torch.Generator().manual_seed(1)
    present = [(0, 1, 2, 3), (0, 1, 3), (0, 1, 3), (0, 1, 2, 3), (1, 2), (0, 1, 2, 3)]      # tensor 2 joins late twice, 0 and 3 skip one
    for it, who in enumerate(present):
        for i, (p, q) in enumerate(zip(a, b)):
            if i in who:
                gr = torch.randn(p.shape, generator=gen)
                p.grad, q.grad = gr.clone(), gr.clone()
            else:
                p.grad = q.grad = None
        hip.param_groups[0]['lr'] = ref.param_groups[0]['lr'] = 1e-2 * (1 - it / 10)       # a schedule
        fake.ntensors = len(who)
        hip.step()
        ref.step()
        for p, q in zip(a, b):
            assert torch.allclose(p, q, rtol=2e-6, atol=2e-7)
    assert [c[0] for c in fake.calls] == [1, 2, 3, 4, 5, 6]                                   # the global step count handed to the kernel
    sd_h, sd_r = hip.state_dict(), ref.state_dict()
    for k in sd_r['state']:
        assert float(sd_h['state'][k]['step']) == float(sd_r['state'][k]['step'])             # torch counts steps per tensor
        assert torch.allclose(sd_h['state'][k]['exp_avg'], sd_r['state'][k]['exp_avg'], rtol=1e-5, atol=1e-6)
        assert torch.allclose(sd_h['state'][k]['exp_avg_sq'], sd_r['state'][k]['exp_avg_sq'], rtol=1e-5, atol=1e-8)


def test_resume_from_a_torch_state_dict_keeps_per_tensor_counts(fake):
    import copy
    a, b = _params(3), _params(3)
    ref = torch.optim.AdamW(_groups(b), foreach=False)
    gen = torch.Generator().manual_seed(5)
    for who in [(0, 1, 2, 3), (0, 1, 3)]:
        for i, q in enumerate(b):
            q.grad = torch.randn(q.shape, generator=gen) if i in who else None
        ref.step()
    for p, q in zip(a, b):
        p.data.copy_(q.data)
    hip = sd_optim.HipAdamW(_groups(a))
    hip.load_state_dict(copy.deepcopy(ref.state_dict()))
    for p, q in zip(a, b):
        gr = torch.randn(p.shape, generator=gen)
        p.grad, q.grad = gr.clone(), gr.clone()
    fake.ntensors = 4
    hip.step()
    ref.step()
    for p, q in zip(a, b):
        assert torch.allclose(p, q, rtol=2e-6, atol=2e-7)
    assert fake.calls[-1][0] == 3                                                              # global step = the largest per-tensor count + 1


def test_unsupported_settings_fail_loudly(fake):
    ps = _params(7)
    opt = sd_optim.HipAdamW([dict(params=ps[:2], betas=(0.9, 0.999)), dict(params=ps[2:], betas=(0.8, 0.999))])
    for p in ps:
        p.grad = torch.zeros_like(p)
    fake.ntensors = 4
    with pytest.raises(RuntimeError, match='betas'):
        opt.step()
