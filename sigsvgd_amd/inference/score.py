"""Score estimator wiring cost -> grad log p and kernel -> (K, grad_k) for SVGD
(reference src/inference/score.py:6-76; only the signature-kernel branch is in scope)."""
from __future__ import annotations

import torch
from torch.autograd import grad as ag

from ..kernels import BaseKernel, SignatureKernel


class ScoreEstimator:
    def __init__(self, kernel, cost_fn, cost_fn_params, scheduler=None, ctx={"device": "cpu"}):
        self.ctx = ctx
        self.kernel = kernel
        self.cost_fn = cost_fn
        self.cost_fn_params = cost_fn_params
        self.scheduler = (lambda: 1) if scheduler is None else scheduler
        if isinstance(self.kernel, SignatureKernel):
            self.score = self._pathsig_score
        elif isinstance(self.kernel, BaseKernel):
            self.score = self._svgd_score if self.kernel.analytic_grad is True else self._svgd_ag_score

    def sgd_score(self, x):
        """No interaction between particles: identity Gram matrix, zero repulsion."""
        cost, cost_dict = self.cost_fn(x, **self.cost_fn_params)
        grad_log_p = ag(-cost.sum(), x, retain_graph=True)[0]  # likelihood is exp(-cost)
        k_xx = torch.eye(x.shape[0], **self.ctx)
        grad_k = torch.zeros_like(grad_log_p, **self.ctx)
        return grad_log_p, {"k_xx": k_xx, "grad_k": grad_k, "loss": cost, **cost_dict}

    def _svgd_score(self, x):
        cost, cost_dict = self.cost_fn(x, **self.cost_fn_params)
        grad_log_p = ag(-cost.sum(), x, retain_graph=True)[0]
        k_xx, grad_k = self.kernel(x, x, compute_grad=True)
        return grad_log_p, {"k_xx": k_xx, "grad_k": self.scheduler() * grad_k, "loss": cost, **cost_dict}

    def _svgd_ag_score(self, x):
        cost, cost_dict = self.cost_fn(x, **self.cost_fn_params)
        grad_log_p = ag(-cost.sum(), x, retain_graph=True)[0]
        k_xx = self.kernel(x, x.detach(), compute_grad=False)
        grad_k = ag(k_xx.sum(), x)[0]
        return grad_log_p, {"k_xx": k_xx, "grad_k": self.scheduler() * grad_k, "loss": cost, **cost_dict}

    def _pathsig_score(self, x):
        """k_xx = K(x, x.detach()), grad_k = d sum(k_xx)/dx -- fused into one HIP launch."""
        cost, cost_dict = self.cost_fn(x, **self.cost_fn_params)
        grad_log_p = ag(-cost.sum(), x, retain_graph=True)[0]
        k_xx, grad_k = self.kernel.gram_and_grad(x.detach())
        score_dict = {"k_xx": k_xx, "grad_k": self.scheduler() * grad_k, "loss": cost, **cost_dict}
        return grad_log_p, score_dict
