"""MPC variant of SVGD: the kernel acts on rolled-out positions and the gradient is chained back to
the sampled actions through the rollout graph (reference src/inference/trajectory_svgd.py:12-84)."""
from __future__ import annotations

from typing import Callable, Tuple

import torch
import torch.autograd as autograd
import torch.optim as optim

from ..kernels import PathSigKernel, TrajectoryKernel
from ..sigkernel import SigKernel
from .svgd import SVGD


class TrajectorySVGD(SVGD):
    def __init__(
        self,
        kernel,
        log_p: Callable = None,
        log_prior: Callable = None,
        bw_scale: float = 1.0,
        gradient_mask=None,
        optimizer_class: optim.Optimizer = optim.Adam,
        **opt_args,
    ):
        super().__init__(kernel, log_p, log_prior, bw_scale, optimizer_class, **opt_args)
        self.gradient_mask = gradient_mask

    def _compute_kernel(self, X, **kwargs):
        if isinstance(self.kernel, TrajectoryKernel):
            tau = kwargs["trajectories"][..., 1:, :2]
            if kwargs["sample_shape"]:
                tau = tau.mean(0)
            k_xx, grad_k = (0, 0)
            for i in range(tau.shape[-1]):
                # gradients w.r.t. the policies = sum of the gradients w.r.t. the sampled actions
                k_xx_i, grad_k_i = self.kernel(tau[..., i], tau[..., i].detach(), X, compute_grad=True)
                k_xx = k_xx + k_xx_i
                grad_k = grad_k + grad_k_i
            k_xx = k_xx / tau.shape[-1]
            grad_k = grad_k.flatten(1) / tau.shape[-1]
            return k_xx.detach(), grad_k.detach()
        if isinstance(self.kernel, PathSigKernel):
            tau = kwargs["trajectories"][..., 1:, :2]
            if kwargs["sample_shape"]:
                tau = tau.mean(0)
            k_xx, grad_k = self.kernel(tau, tau, X, compute_grad=True)
            return k_xx.detach(), grad_k.detach().flatten(1)
        if isinstance(self.kernel, SigKernel):
            # x, y positions from time t+1 on; mean over action samples if present
            tau = kwargs["trajectories"][..., 1:, :2]
            if kwargs["sample_shape"]:
                tau = tau.mean(0)
            # (the reference upcasts to fp64 here and casts K back to fp32; the HIP kernel computes
            #  in fp64 from the fp32 paths directly)
            k_xx = self.kernel.compute_Gram(tau, tau.detach(), sym=False).float()
            grad_k = torch.autograd.grad(k_xx.sum(), kwargs["actions"])[0]
            if kwargs["sample_shape"]:
                grad_k = grad_k.mean(0).flatten(1)
            else:
                grad_k = grad_k.flatten(1)
            return k_xx.detach(), grad_k.detach()
        if hasattr(self.kernel, "analytic_grad") and self.kernel.analytic_grad:
            k_xx, grad_k = self.kernel(X, X)
            grad_k = grad_k.sum(1)  # aggregate the gradient w.r.t. the first input
        else:
            X = X.detach().requires_grad_(True)
            k_xx = self.kernel(X, X.detach(), compute_grad=False)
            grad_k = autograd.grad(-k_xx.sum(), X)[0]
        return k_xx.detach(), grad_k.detach()

    def _velocity(self, X: torch.Tensor, grad_log_p: torch.Tensor, **kwargs) -> Tuple[torch.Tensor, dict]:
        velocity, iter_dict = super()._velocity(X, grad_log_p, **kwargs)
        return velocity * self.gradient_mask, iter_dict
