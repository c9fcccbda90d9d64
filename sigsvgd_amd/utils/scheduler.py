"""Scalar schedules that multiply the repulsive term grad_k (reference API: src/utils/scheduler.py:4,25,50,
consumed at src/inference/score.py:72).  A schedule is a callable; every call returns the value for the
current epoch and, unless `update_epoch=False`, moves on to the next one.

All three share one stepping mechanism (`_Schedule`) and differ only in `value_at(epoch)`; attribute names
(`param`, `last_epoch`, ...) follow the reference because user code reads and resets them.
"""
from __future__ import annotations

import math

import torch


class _Schedule:
    """Epoch counter + call protocol; subclasses provide value_at(epoch) -> 0-dim tensor."""

    def __init__(self, parameter):
        self.param = torch.as_tensor(parameter)
        self.last_epoch = 0

    def value_at(self, epoch: int) -> torch.Tensor:  # pragma: no cover - abstract
        raise NotImplementedError

    def __call__(self, update_epoch: bool = True) -> torch.Tensor:
        epoch = self.last_epoch
        self.last_epoch = epoch + (1 if update_epoch else 0)
        return self.value_at(epoch)


class SquareRootScheduler(_Schedule):
    r"""rho_t = rho_0 / sqrt(t + 1)."""

    def value_at(self, epoch):
        return self.param * (epoch + 1) ** -0.5


class FactorScheduler(_Schedule):
    r"""rho_t = max(rho_min, rho_0 * gamma^t)."""

    def __init__(self, parameter, gamma, parameter_min=1e-7):
        super().__init__(parameter)
        self.gamma = gamma
        self.param_min = torch.as_tensor(parameter_min)

    def value_at(self, epoch):
        return torch.max(self.param_min, self.param * self.gamma**epoch)


class CosineScheduler(_Schedule):
    r"""rho_0 while t <= warmup_steps; rho_T after final_epoch; in between the half-cosine
    rho_T + (rho_0 - rho_T)/2 * (1 + cos(pi (t - warmup_steps) / final_epoch)) -- the phase is divided by
    final_epoch, not by the length of the decay window, as in the reference."""

    def __init__(self, parameter, target_paremeter, final_epoch, warmup_steps=0):
        super().__init__(parameter)
        self.target = torch.as_tensor(target_paremeter)  # (the keyword keeps the reference's spelling)
        self.final_epoch = final_epoch
        self.warmup = warmup_steps
        self.pi = torch.tensor(math.pi)

    def value_at(self, epoch):
        if epoch <= self.warmup:
            return self.param
        if epoch > self.final_epoch:
            return self.target
        swing = 1 + torch.cos(self.pi * (epoch - self.warmup) / self.final_epoch)
        return self.target + (self.param - self.target) / 2 * swing
