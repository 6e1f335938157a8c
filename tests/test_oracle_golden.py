"""The CPU oracle (oracle/kd_ref.py) against outputs of the reference itself
(tests/golden/kd_losses.npz, produced by oracle/gen_golden.py from
/root/reference/mmseg/models/distillation/losses.py)."""
import numpy as np
import pytest
import torch

from oracle import kd_ref
from oracle.inputs import kat_pair, wavy_pair, probe_vector


def _eager(s, t, hw, **kw):
    xs = torch.tensor(np.asarray(s), dtype=torch.float64, requires_grad=True)
    loss = kd_ref.eager_kld(xs, torch.tensor(np.asarray(t), dtype=torch.float64), out_size=hw, **kw)
    (g,) = torch.autograd.grad(loss, xs)
    return float(loss), g.numpy()


G1 = {
    'cgd_g4_a3_t4': dict(hw=(8, 8), alpha=3, tau=4, group_size=4),
    'cgd_g3_a3_t2': dict(hw=(8, 8), alpha=3, tau=2, group_size=3),
    'cgd_g6_a1_t1': dict(hw=(8, 8), alpha=1, tau=1, group_size=6),
    'cd': dict(hw=(8, 8), alpha=1, tau=1, group_size=1),
    'pd': dict(hw=(8, 8), alpha=1, tau=1, loss_type='pixel', group_size=None),
    'kld_g2_a2_t3_noresize': dict(hw=None, alpha=2, tau=3, group_size=2),
}


def test_survey_known_answers(golden):
    # values quoted in SURVEY.md section 8(c) G1, independently of the npz
    assert abs(float(golden['G1/cgd_g4_a3_t4/loss']) - 0.503972902723) < 1e-11
    assert abs(float(golden['G1/cd/loss']) - 2.160378509155) < 1e-11
    assert abs(float(golden['G1/pd/loss']) - 0.657146405165) < 1e-11
    assert abs(float(golden['G1/at/loss']) - 6.122915720687) < 1e-11


@pytest.mark.parametrize('name', sorted(G1))
def test_g1_eager_and_rowwise(golden, name):
    kw = dict(G1[name])
    hw = kw.pop('hw')
    s, t = kat_pair()
    loss, grad = _eager(s, t, hw, **kw)
    assert abs(loss - float(golden[f'G1/{name}/loss'])) < 1e-12
    np.testing.assert_allclose(grad, golden[f'G1/{name}/grad'], rtol=0, atol=1e-14)
    full = kd_ref.full_kld(s, t, out_size=hw, **kw)
    assert abs(full['loss'] - float(golden[f'G1/{name}/loss'])) < 1e-12
    np.testing.assert_allclose(full['grad_s'], golden[f'G1/{name}/grad'], rtol=0, atol=1e-13)


def test_g1_at_ifvd(golden):
    s, t = kat_pair()
    xs = torch.tensor(s, requires_grad=True)
    loss = kd_ref.eager_at(xs, torch.tensor(t))
    (g,) = torch.autograd.grad(loss, xs)
    assert abs(float(loss) - float(golden['G1/at/loss'])) < 1e-12
    np.testing.assert_allclose(g.numpy(), golden['G1/at/grad'], atol=1e-13)
    xs = torch.tensor(s, requires_grad=True)
    loss = kd_ref.eager_ifvd(xs, torch.tensor(t), torch.tensor(golden['G1/ifvd/label']))
    (g,) = torch.autograd.grad(loss, xs)
    assert abs(float(loss) - float(golden['G1/ifvd/loss'])) < 1e-11
    np.testing.assert_allclose(g.numpy(), golden['G1/ifvd/grad'], atol=1e-12)


def test_alpha_schedules(golden):
    s, t = kat_pair()
    base = _eager(s, t, (8, 8), alpha=1.0, tau=2, group_size=10)[0]
    sch = kd_ref.AlphaSchedule(3, kd_ref.PRESETS['CGDLossWS']['warmup'], kd_ref.PRESETS['CGDLossWS']['earlydecay'])
    for it, loss, alpha in zip(golden['G1/cgdws_trace/iters'], golden['G1/cgdws_trace/loss'],
                               golden['G1/cgdws_trace/alpha']):
        a = sch.step(int(it))
        assert a == pytest.approx(float(alpha), abs=1e-15), it
        assert a * base == pytest.approx(float(loss), abs=1e-12), it
    for wm in ('linear', 'exp', 'jump'):
        for dm in ('linear', 'exp', 'jump'):
            sch = kd_ref.AlphaSchedule(2.5, {'mode': wm, 'warmup_iters': 10},
                                       {'mode': dm, 'earlydecay_start': 20, 'earlydecay_end': 30})
            key = f'G1/sched_{wm}_{dm}'
            for it, alpha in zip(golden[key + '/iters'], golden[key + '/alpha']):
                assert sch.step(int(it)) == pytest.approx(float(alpha), abs=1e-15), (wm, dm, it)


def test_g2_g3(golden):
    s, t = golden['G2/inputs/s'], golden['G2/inputs/t']
    s2, t2 = wavy_pair((2, 22, 16, 16))
    assert np.array_equal(s, s2) and np.array_equal(t, t2)
    for key in [k[:-5] for k in golden.files if k.startswith('G2/g') and k.endswith('/loss')]:
        g, a, tau = golden[key + '/cfg']
        full = kd_ref.full_kld(s, t, out_size=(64, 64), alpha=a, tau=tau, group_size=int(g))
        assert full['loss'] == pytest.approx(float(golden[key + '/loss']), rel=1e-12)
        np.testing.assert_allclose(full['grad_s'], golden[key + '/grad'], atol=1e-13)
        loss, grad = _eager(s, t, (64, 64), alpha=a, tau=tau, group_size=int(g))
        assert loss == pytest.approx(float(golden[key + '/loss']), rel=1e-13)
    full = kd_ref.full_kld(s, t, out_size=(64, 64), alpha=1, tau=1, loss_type='pixel')
    assert full['loss'] == pytest.approx(float(golden['G2/pd/loss']), rel=1e-12)
    np.testing.assert_allclose(full['grad_s'], golden['G2/pd/grad'], atol=1e-13)
    for hw in [(40, 56), (16, 16), (24, 100)]:
        key = f'G2/resize_{hw[0]}x{hw[1]}'
        full = kd_ref.full_kld(s, t, out_size=hw, alpha=3, tau=4, group_size=8)
        assert full['loss'] == pytest.approx(float(golden[key + '/loss']), rel=1e-12)
        np.testing.assert_allclose(full['grad_s'], golden[key + '/grad'], atol=1e-13)
    for seed in (0, 7):
        key = f'G3/seed{seed}'
        full = kd_ref.full_kld(s, t, out_size=(64, 64), alpha=3, tau=4, group_size=8, perm=golden[key + '/perm'])
        assert full['loss'] == pytest.approx(float(golden[key + '/loss']), rel=1e-12)
        np.testing.assert_allclose(full['grad_s'], golden[key + '/grad'], atol=1e-13)
        loss, grad = _eager(s, t, (64, 64), alpha=3, tau=4, group_size=8, perm=golden[key + '/perm'])
        np.testing.assert_allclose(grad, golden[key + '/grad'], atol=1e-14)


def test_g4_ade_shaped(golden):
    key = 'G4/ade32_g8_a3_t4'
    shp = tuple(int(v) for v in golden[key + '/shape'])
    s, t = wavy_pair(shp)
    loss, grad = _eager(s, t, (128, 128), alpha=3, tau=4, group_size=8)
    assert loss == pytest.approx(float(golden[key + '/loss64']), rel=1e-12)
    assert (grad * probe_vector(shp)).sum() == pytest.approx(float(golden[key + '/grad_probe']), rel=1e-9)
    assert np.abs(grad).sum() == pytest.approx(float(golden[key + '/grad_abs_sum']), rel=1e-11)
    np.testing.assert_allclose(grad.reshape(-1)[::4099][:64], golden[key + '/grad_sample'], atol=1e-15)
