"""TEST INFRASTRUCTURE -- CPU restatement of the reference's planning cost, the checker for the HIP cost kernel.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Follows examples/script_planning_obstacle_field.py of the reference:
    :18-23    create_spline_trajectory -- natural cubic spline (torchcubicspline, a third-party package absent
              from the reference tree and this image) through knots at uniform times in [0, 1], sampled at
              `timesteps` uniform times.  Restated with scipy.interpolate.CubicSpline(bc_type="natural"): the
              natural cubic spline through given knots is unique, so the two agree to rounding.
    :113-126  batch_cost_fn -- knots = [start, x, target]; obstacle cost = sum_t w0 * exp(log_p(traj_t));
              length cost = Frobenius norm of w1 * (traj[1:] - traj[:-1]).
    :363-370  the obstacle field log_p = MixtureSameFamily(Categorical(w), Independent(Normal(mean, std), 1)).log_prob,
              built here with the same torch.distributions classes.
Gradients come from torch autograd in fp64, as the reference's ScoreEstimator takes them
(src/inference/_likelihoods.py: grad of -cost.sum() w.r.t. x).

Parity pin: the reference's cost is a closure inside run_exp() (not importable) and its spline dependency is absent,
so there is no reference run to pin against; the pin is the identity of the torch.distributions classes, the
uniqueness of the natural spline, and finite differences of this restatement (tests/test_cost_oracle.py).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributions as dist
from scipy.interpolate import CubicSpline


def spline_samples_matrix(n_knots: int, timesteps: int) -> torch.Tensor:
    """[timesteps, n_knots] fp64 matrix of create_spline_trajectory (:18-23) for unit knot vectors."""
    t_knots = np.linspace(0.0, 1.0, n_knots)
    t = np.linspace(0.0, 1.0, timesteps)
    if n_knots == 2:  # a natural spline through two points is the chord
        return torch.as_tensor(np.stack([1.0 - t, t], axis=1))
    return torch.as_tensor(CubicSpline(t_knots, np.eye(n_knots), bc_type="natural")(t))


def obstacle_field(weights: torch.Tensor, mean: torch.Tensor, std: torch.Tensor):
    """:363-370 -- returns the log_prob callable the script hands to the cost function."""
    mix = dist.Categorical(weights.double())
    comp = dist.Independent(dist.Normal(mean.double(), std.double()), 1)
    return dist.MixtureSameFamily(mix, comp).log_prob


def batch_cost_fn(x, log_p, start_pose, target_pose, timesteps=100, w=(1.0, 1.0), use_splines=True):
    """:113-126 in fp64.  x [batch, knots, d] -> (cost [batch], trajectories [batch, timesteps, d])."""
    x = x.double()
    batch = x.shape[0]
    knots = torch.cat((start_pose.double().reshape(1, 1, -1).repeat(batch, 1, 1), x,
                       target_pose.double().reshape(1, 1, -1).repeat(batch, 1, 1)), 1)
    traj = spline_samples_matrix(knots.shape[1], timesteps) @ knots if use_splines else knots
    obst_cost = (w[0] * log_p(traj).exp()).sum(-1)
    len_cost = torch.norm(w[1] * (traj[:, 1:] - traj[:, :-1]), dim=[-2, -1])
    return obst_cost + len_cost, traj


def cost_and_grad(x, weights, mean, std, start_pose, target_pose, timesteps=100, w=(1.0, 1.0), use_splines=True):
    """(cost [batch], traj, d cost / d x) as fp64 numpy arrays."""
    xr = x.detach().double().clone().requires_grad_(True)
    cost, traj = batch_cost_fn(xr, obstacle_field(weights, mean, std), start_pose, target_pose, timesteps, w, use_splines)
    (g,) = torch.autograd.grad(cost.sum(), xr)
    return cost.detach().numpy(), traj.detach().numpy(), g.numpy()
