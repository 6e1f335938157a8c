"""Bookkeeping of the deferred weight-gradient scope's held-operand budget (segdistill_amd/deferred.py) without a GPU: storages are
counted once, the budget triggers the early launch, the count returns to zero at the scope's end."""
import torch

from segdistill_amd import deferred


def test_budget_counts_storages_once_and_triggers_the_early_flush(monkeypatch):
    calls = []
    monkeypatch.setattr(deferred, '_flush_wgrads', lambda: (calls.append(len(deferred._wgrads)), deferred._wgrads.clear()))
    monkeypatch.setattr(deferred, '_HELD_BUDGET_MB', str(3 * 4096 * 4 / (1 << 20)))       # three 16 KB storages
    monkeypatch.setattr(deferred, 'flush', lambda: deferred.flush_operands())
    base = deferred.partial_flushes
    with deferred.scope():
        x = torch.zeros(64, 64)
        for k in range(2):
            dy = torch.zeros(64, 64)
            deferred._wgrads.append((dy, x))
            deferred._note_held(dy, x)                    # x is one storage however often it is registered
        assert deferred.held_bytes() == 3 * 64 * 64 * 4 and not calls          # dy0, dy1, x: at the budget, not over it
        dy = torch.zeros(64, 64)
        deferred._wgrads.append((dy, x))
        deferred._note_held(dy, x)                        # fourth storage: over -> the three registered products run now
        assert calls == [3] and deferred.held_bytes() == 0
        dy = torch.zeros(64, 64)
        deferred._wgrads.append((dy, x))
        deferred._note_held(dy, x)
        assert deferred.held_bytes() == 2 * 64 * 64 * 4
    assert calls == [3, 1] and deferred.partial_flushes == base + 1
    assert deferred.held_bytes() == 0


def test_default_budget_is_a_fraction_of_the_device_or_unbounded_without_one():
    assert deferred.held_budget_bytes() > (1 << 30)
