"""Scalar schedules multiplying the repulsive term grad_k (reference src/utils/scheduler.py:4,25,50;
used at src/inference/score.py:72).  Each call returns the current value and advances the epoch."""
from __future__ import annotations

import math

import torch


class SquareRootScheduler:
    r"""rho_t = rho_0 (t+1)^{-1/2}."""

    def __init__(self, parameter):
        self.param = torch.as_tensor(parameter)
        self.last_epoch = 0

    def __call__(self, update_epoch=True):
        val = self.param * (self.last_epoch + 1) ** -0.5
        if update_epoch:
            self.last_epoch += 1
        return val


class FactorScheduler:
    r"""rho_t = max(rho_min, rho_0 * gamma^t)."""

    def __init__(self, parameter, gamma, parameter_min=1e-7):
        self.param = torch.as_tensor(parameter)
        self.gamma = gamma
        self.param_min = torch.as_tensor(parameter_min)
        self.last_epoch = 0

    def __call__(self, update_epoch=True):
        val = torch.max(self.param_min, self.param * self.gamma**self.last_epoch)
        if update_epoch:
            self.last_epoch += 1
        return val


class CosineScheduler:
    r"""Constant until `warmup_steps`, then rho_T + (rho_0-rho_T)/2 (1 + cos(pi (t - warmup)/T))
    while t <= T = final_epoch, then rho_T.  (The denominator is final_epoch, as in the reference.)"""

    def __init__(self, parameter, target_paremeter, final_epoch, warmup_steps=0):
        self.param = torch.as_tensor(parameter)
        self.target = torch.as_tensor(target_paremeter)
        self.final_epoch = final_epoch
        self.warmup = warmup_steps
        self.last_epoch = 0
        self.pi = torch.tensor(math.pi)

    def __call__(self, update_epoch=True):
        t = self.last_epoch
        if t <= self.warmup:
            val = self.param
        elif t <= self.final_epoch:
            phase = torch.cos(self.pi * (t - self.warmup) / self.final_epoch)
            val = self.target + (self.param - self.target) / 2 * (1 + phase)
        else:
            val = self.target
        if update_epoch:
            self.last_epoch += 1
        return val
