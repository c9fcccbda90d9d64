"""Cost functions in front of the signature-kernel path, on the device.

`ObstacleFieldCost` is the cost of the reference's planning script (examples/script_planning_obstacle_field.py:113-126,
`batch_cost_fn`) for the obstacle field that script builds (:363-370, a mixture of axis-aligned Gaussians): it has
the call shape ScoreEstimator expects of a `cost_fn` -- `cost, aux = cost_fn(x, **params)` with
`aux["trajectories"]` -- and is differentiable, so `ScoreEstimator.grad_log_p` works unchanged; but the forward is
ONE HIP launch (spline samples, field, length) that also produces d cost / d x analytically, and backward only scales
it.  The reference's version builds a torch graph of ~20 small ops per call and differentiates it with autograd.
"""
from __future__ import annotations

from typing import Sequence

import torch

from . import ops
from .utils.spline import spline_basis


class _ObstacleCostFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, start, target, basis, log_weights, mean, std, w0, w1):
        cost, traj, grad = ops.obstacle_cost(x, start, target, basis, log_weights, mean, std, w0, w1)
        ctx.save_for_backward(grad)
        ctx.mark_non_differentiable(traj)
        return cost.to(x.dtype), traj.to(x.dtype)

    @staticmethod
    def backward(ctx, g_cost, _g_traj):
        (grad,) = ctx.saved_tensors
        return (g_cost.reshape(-1, 1, 1).to(grad.dtype) * grad,) + (None,) * 8


class ObstacleFieldCost:
    """cost_fn(x) -> (cost [batch], {"trajectories": [batch, timesteps, d]}) for x [batch, knots, d] on the GPU.

    weights/mean/std describe the field exactly as the script's
    MixtureSameFamily(Categorical(weights), Independent(Normal(mean, std), 1)); `w` = [w_obstacle, w_length];
    use_splines=False treats [start, x, target] itself as the trajectory (script :121-124)."""

    def __init__(self, weights: torch.Tensor, mean: torch.Tensor, std: torch.Tensor, start_pose: torch.Tensor,
                 target_pose: torch.Tensor, timesteps: int = 100, w: Sequence[float] = (1.0, 1.0),
                 use_splines: bool = True):
        dev = mean.device
        wts = weights.detach().double().reshape(-1)
        if bool((wts < 0).any()) or float(wts.sum()) <= 0:
            raise ValueError("mixture weights must be non-negative with a positive sum")
        self.log_weights = (wts / wts.sum()).log().to(device=dev, dtype=torch.float32)
        self.mean = mean.detach().to(torch.float32).contiguous()
        self.std = std.detach().to(device=dev, dtype=torch.float32).contiguous()
        if bool((self.std <= 0).any()):
            raise ValueError("component standard deviations must be positive")
        self.start = start_pose.detach().to(device=dev, dtype=torch.float32).reshape(-1)
        self.target = target_pose.detach().to(device=dev, dtype=torch.float32).reshape(-1)
        self.timesteps, self.w, self.use_splines = int(timesteps), (float(w[0]), float(w[1])), bool(use_splines)
        self._basis = {}

    def basis(self, n_knots: int) -> torch.Tensor:
        B = self._basis.get(n_knots)
        if B is None:
            if self.use_splines:
                B = spline_basis(torch.linspace(0, 1, n_knots), torch.linspace(0, 1, self.timesteps))
            else:
                B = torch.eye(n_knots, dtype=torch.float64)
            B = self._basis[n_knots] = B.to(device=self.mean.device, dtype=torch.float32).contiguous()
        return B

    def __call__(self, x: torch.Tensor, w: Sequence[float] = None, **_script_params):
        """Accepts (and ignores) the script's other cost_fn_params -- log_p, start_pose, target_pose, timesteps --
        which this object was constructed from; `w` overrides the weights per call as in the script."""
        w0, w1 = self.w if w is None else (float(w[0]), float(w[1]))
        ts = _script_params.get("timesteps")
        if ts is not None and self.use_splines and int(ts) != self.timesteps:
            raise ValueError(f"this cost was built for {self.timesteps} timesteps, called with {ts}")
        cost, traj = _ObstacleCostFn.apply(x, self.start, self.target, self.basis(x.shape[1] + 2), self.log_weights,
                                           self.mean, self.std, w0, w1)
        return cost, {"trajectories": traj}

    def cost_and_score(self, x: torch.Tensor):
        """(cost, trajectories, grad log p = -d cost / d x) straight from the kernel, without an autograd graph."""
        cost, traj, grad = ops.obstacle_cost(x, self.start, self.target, self.basis(x.shape[1] + 2), self.log_weights,
                                             self.mean, self.std, self.w[0], self.w[1])
        return cost, traj, grad.neg_()
