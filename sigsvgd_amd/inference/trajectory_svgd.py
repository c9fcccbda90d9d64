"""SVGD over control sequences for the MPC loop: the kernel compares rolled-out POSITIONS and its gradient is
chained back to the sampled actions through the rollout graph (reference API:
src/inference/trajectory_svgd.py:12-84).  `TrajectorySVGD(kernel, ..., gradient_mask=...)`; the step receives
`trajectories=[..., horizon+1, state]`, `actions=...` and `sample_shape=...` as keyword arguments.
"""
from __future__ import annotations

from typing import Callable, Tuple

import torch
import torch.optim as optim

from ..kernels import PathSigKernel, TrajectoryKernel
from ..sigkernel import SigKernel
from .svgd import SVGD


def _xy_positions(kwargs) -> torch.Tensor:
    """x, y positions of the rollouts from time t+1 on, averaged over the action samples if there are any."""
    tau = kwargs["trajectories"][..., 1:, :2]
    return tau.mean(0) if kwargs["sample_shape"] else tau


class TrajectorySVGD(SVGD):
    def __init__(self, kernel, log_p: Callable = None, log_prior: Callable = None, bw_scale: float = 1.0,
                 gradient_mask=None, optimizer_class: optim.Optimizer = optim.Adam, **opt_args):
        super().__init__(kernel, log_p, log_prior, bw_scale, optimizer_class, **opt_args)
        self.gradient_mask = gradient_mask

    # one method per kernel family; each returns (k_xx, grad_k) still attached where the reference's is
    def _kernel_per_coordinate(self, X, kwargs):
        tau = _xy_positions(kwargs)
        ncoord = tau.shape[-1]
        k_sum, g_sum = 0, 0
        for c in range(ncoord):  # the gradient w.r.t. the policy is the sum over the sampled actions
            k_c, g_c = self.kernel(tau[..., c], tau[..., c].detach(), X, compute_grad=True)
            k_sum, g_sum = k_sum + k_c, g_sum + g_c
        return k_sum / ncoord, g_sum.flatten(1) / ncoord

    def _kernel_truncated_signature(self, X, kwargs):
        tau = _xy_positions(kwargs)
        k_xx, grad_k = self.kernel(tau, tau, X, compute_grad=True)
        return k_xx, grad_k.detach().flatten(1)

    def _kernel_signature_pde(self, X, kwargs):
        tau = _xy_positions(kwargs)
        # (the reference upcasts to fp64 around compute_Gram and casts K back to fp32; the HIP kernels compute in
        #  fp64 from the fp32 paths directly)
        k_xx = self.kernel.compute_Gram(tau, tau.detach(), sym=False).float()
        (grad_k,) = torch.autograd.grad(k_xx.sum(), kwargs["actions"])
        if kwargs["sample_shape"]:
            grad_k = grad_k.mean(0)
        return k_xx, grad_k.flatten(1)

    def _kernel_on_particles(self, X):
        if getattr(self.kernel, "analytic_grad", False):
            k_xx, grad_k = self.kernel(X, X)
            return k_xx, grad_k.sum(1)  # aggregate over the first input, as the reference does here
        Xg = X.detach().requires_grad_(True)
        k_xx = self.kernel(Xg, Xg.detach(), compute_grad=False)
        (grad_k,) = torch.autograd.grad(-k_xx.sum(), Xg)
        return k_xx, grad_k

    def _compute_kernel(self, X, **kwargs):
        if isinstance(self.kernel, TrajectoryKernel):
            k_xx, grad_k = self._kernel_per_coordinate(X, kwargs)
        elif isinstance(self.kernel, PathSigKernel):
            k_xx, grad_k = self._kernel_truncated_signature(X, kwargs)
        elif isinstance(self.kernel, SigKernel):
            k_xx, grad_k = self._kernel_signature_pde(X, kwargs)
        else:
            k_xx, grad_k = self._kernel_on_particles(X)
        return k_xx.detach(), grad_k.detach()

    def _velocity(self, X: torch.Tensor, grad_log_p: torch.Tensor, **kwargs) -> Tuple[torch.Tensor, dict]:
        velocity, iter_dict = super()._velocity(X, grad_log_p, **kwargs)
        return velocity * self.gradient_mask, iter_dict
