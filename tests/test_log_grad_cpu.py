"""`log_grad` gradient-angle diagnostic (reference mmseg/models/segmentors/SD_structure.py:92-108 get_grads, :124-134): the host logic on a
tiny student, against the reference's own recipe (backward into .grad, concatenate, zero) written out here."""
from collections import OrderedDict

import pytest
import torch
import torch.nn as nn

from segdistill_amd.segmentors.base import parse_losses
from segdistill_amd.segmentors.sd_module import SDModule


class _Student(nn.Module):
    def __init__(self):
        super().__init__()
        self.body = nn.Linear(6, 5)
        self.decode = nn.Linear(5, 3)
        self.aux = nn.Linear(5, 3)       # reached by the segmentation loss only
        self.frozen = nn.Linear(5, 3)
        for p in self.frozen.parameters():
            p.requires_grad = False


def _module(student):
    m = SDModule.__new__(SDModule)
    nn.Module.__init__(m)
    m.student = student
    m.log_grad = True
    m.defer_log_sync = False
    return m


def _losses(student, x):
    h = torch.tanh(student.body(x))
    seg = (student.decode(h) ** 2).mean() + (student.aux(h) ** 2).mean()
    kd = (student.decode(h) - 0.3).abs().mean() + student.frozen(h).mean() * 0
    return OrderedDict([('decode.loss_seg', seg), ('decode.acc_seg', torch.tensor(12.5)), ('loss_channel_kd', kd)])


def _reference_deg(student, losses):
    def get_grads(loss):
        loss.backward(retain_graph=True)
        g = torch.cat([p.grad.flatten().clone() for p in student.parameters() if p.requires_grad and p.grad is not None])
        for p in student.parameters():          # zero_grad() of the reference's torch: grads zeroed in place, tensors kept
            if p.grad is not None:
                p.grad.zero_()
        return g
    a, b = get_grads(losses['decode.loss_seg']), get_grads(losses['loss_channel_kd'])
    for p in student.parameters():
        p.grad = None
    return torch.acos(torch.sum(a * b) / (torch.norm(a) * torch.norm(b))) * 180 / 3.1416


def test_deg_matches_reference_recipe_and_leaves_grads_alone():
    torch.manual_seed(3)
    student = _Student()
    x = torch.randn(7, 6)
    want = float(_reference_deg(student, _losses(student, x)))
    m = _module(student)
    loss, log_vars = m._parse_losses(_losses(student, x))
    assert list(log_vars) == ['decode.loss_seg', 'decode.acc_seg', 'loss_channel_kd', 'deg', 'loss']   # deg in front of loss, as in the reference
    assert log_vars['deg'] == pytest.approx(want, rel=1e-5)
    assert 0.0 < log_vars['deg'] < 180.0
    assert all(p.grad is None for p in student.parameters())      # the step's own backward starts clean
    assert log_vars['loss'] == pytest.approx(log_vars['decode.loss_seg'] + log_vars['loss_channel_kd'], rel=1e-6)   # deg / acc not summed
    loss.backward()                                                # the graph is still alive
    assert student.body.weight.grad is not None


def test_missing_channel_key_fails_like_the_reference():
    student = _Student()
    m = _module(student)
    losses = _losses(student, torch.randn(2, 6))
    losses['loss_KLDLoss'] = losses.pop('loss_channel_kd')
    with pytest.raises(UnboundLocalError):
        m._parse_losses(losses)


def test_parse_losses_without_extra_is_unchanged():
    loss, log_vars = parse_losses(OrderedDict([('a.loss_x', torch.tensor([1.0, 3.0])), ('acc', torch.tensor(5.0))]))
    assert list(log_vars) == ['a.loss_x', 'acc', 'loss'] and log_vars['loss'] == 2.0
