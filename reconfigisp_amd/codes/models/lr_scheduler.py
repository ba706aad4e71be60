"""Learning-rate schedules with warm restarts used by the model wrappers.

Same names / constructor arguments / resulting LR sequences as the reference's
models/lr_scheduler.py:8-62 (``MultiStepLR_Restart``, ``CosineAnnealingLR_Restart``);
host-side only, no device work."""
import math
from collections import Counter

from torch.optim.lr_scheduler import _LRScheduler


class _WithRestarts(_LRScheduler):
    """At iteration restarts[i] the LR jumps back to initial_lr * weights[i]."""

    def _init_restarts(self, restarts, weights):
        self.restarts = list(restarts) if restarts else [0]
        self.restart_weights = list(weights) if weights else [1]
        if len(self.restarts) != len(self.restart_weights):
            raise AssertionError('restarts and their weights do not match.')

    def _restart_index(self):
        return self.restarts.index(self.last_epoch) if self.last_epoch in self.restarts else None

    def _restarted(self, idx):
        scale = self.restart_weights[idx]
        return [group['initial_lr'] * scale for group in self.optimizer.param_groups]

    def _current(self):
        return [group['lr'] for group in self.optimizer.param_groups]


class MultiStepLR_Restart(_WithRestarts):
    def __init__(self, optimizer, milestones, restarts=None, weights=None, gamma=0.1, clear_state=False,
                 last_epoch=-1):
        self.milestones = Counter(milestones)
        self.gamma = gamma
        self.clear_state = clear_state
        self._init_restarts(restarts, weights)
        super().__init__(optimizer, last_epoch)

    def get_lr(self):
        idx = self._restart_index()
        if idx is not None:
            if self.clear_state:
                self.optimizer.state.clear()          # drop momentum / Adam moments at a restart
            return self._restarted(idx)
        hits = self.milestones.get(self.last_epoch, 0)
        return [lr * self.gamma ** hits for lr in self._current()] if hits else self._current()


class CosineAnnealingLR_Restart(_WithRestarts):
    def __init__(self, optimizer, T_period, restarts=None, weights=None, eta_min=0, last_epoch=-1):
        self.T_period = T_period
        self.T_max = T_period[0]
        self.eta_min = eta_min
        self.last_restart = 0
        self._init_restarts(restarts, weights)
        super().__init__(optimizer, last_epoch)

    def get_lr(self):
        if self.last_epoch == 0:
            return self.base_lrs
        idx = self._restart_index()
        if idx is not None:
            self.last_restart, self.T_max = self.last_epoch, self.T_period[idx + 1]
            return self._restarted(idx)
        t = self.last_epoch - self.last_restart
        if (t - 1 - self.T_max) % (2 * self.T_max) == 0:
            bump = (1 - math.cos(math.pi / self.T_max)) / 2
            return [lr + (base - self.eta_min) * bump for base, lr in zip(self.base_lrs, self._current())]
        ratio = (1 + math.cos(math.pi * t / self.T_max)) / (1 + math.cos(math.pi * (t - 1) / self.T_max))
        return [ratio * (lr - self.eta_min) + self.eta_min for lr in self._current()]
