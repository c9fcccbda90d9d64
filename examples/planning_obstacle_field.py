"""The reference's 2-D planning experiment (examples/script_planning_obstacle_field.py, its SigSVGD leg :152-166 with
the hyperparameters of :313-334) with every per-iteration stage on the MI355X:

    knots -> spline samples, obstacle + length cost and grad log p   one HIP launch   (sigsvgd_amd.costs)
    signature-kernel Gram matrix and its repulsive gradient          one HIP launch   (SignatureKernel)
    velocity + Adam update                                           one HIP launch   (SVGD.step, fused)

20 splines of 3 free knots between a fixed start and target pose, a field of Gaussian obstacles at Halton points,
Adam lr 0.05, signature kernel with bandwidth 0.03 and dyadic order 5.  (The committed reference script shifts the
obstacle means by +100, i.e. out of the workspace; `--shift 100` reproduces that, the default keeps them in view.)

    python examples/planning_obstacle_field.py [--steps 100] [--obstacles 10] [--seed 0]
"""
from __future__ import annotations

import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scipy.stats import qmc


def run(steps: int = 100, n_obst: int = 10, seed: int = 0, device: str = "cuda:0", shift: float = 0.0,
        w=(1.0, 1.0)):
    from sigsvgd_amd.costs import ObstacleFieldCost
    from sigsvgd_amd.inference import SVGD, ScoreEstimator
    from sigsvgd_amd.kernels import SignatureKernel

    dev = torch.device(device)
    ctx = {"device": dev, "dtype": torch.float32}
    batch, length = 20, 5
    gen = torch.Generator().manual_seed(seed)
    start_pose, target_pose = torch.tensor([0.25, 0.75], **ctx), torch.tensor([4.75, 4.5], **ctx)
    xs = torch.linspace(0.25, 4.75, length) + 0.4 * torch.randn(batch, length, generator=gen)
    ys = torch.linspace(0.75, 4.5, length) + 0.4 * torch.randn(batch, length, generator=gen)
    x0 = torch.stack([xs[:, 1:-1], ys[:, 1:-1]], dim=-1).to(**ctx)
    # the field of :363-370
    limits = torch.tensor([[0.0, 0.0], [5.0, 5.0]])
    mean = qmc.scale(qmc.Halton(2, seed=seed).random(n_obst), (limits[0] + 0.5).numpy(), (limits[1] - 0.5).numpy())
    mean = torch.as_tensor(mean, **ctx) + shift
    cost_fn = ObstacleFieldCost(torch.ones(n_obst, **ctx), mean, 0.05 * torch.ones(n_obst, 2, **ctx), start_pose,
                                target_pose, timesteps=100, w=w)
    kernel = SignatureKernel(bandwidth_fn=lambda _: 0.03, depth=5)
    sampler = SVGD(kernel, optimizer_class=torch.optim.Adam, adaptive_gradient=True, lr=0.05)
    estimator = ScoreEstimator(kernel, cost_fn, {"w": list(w)}, None, ctx)
    particles = x0.clone()
    cost0 = cost_fn(particles)[0]
    sampler.optimize(particles, estimator.score, n_steps=steps)
    cost1, aux = cost_fn(particles)
    return {"cost_initial": float(cost0.mean()), "cost_final": float(cost1.mean()), "best_final": float(cost1.min()),
            "trajectories": aux["trajectories"], "knots": particles.detach()}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--obstacles", type=int, default=10)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--shift", type=float, default=0.0)
    a = ap.parse_args()
    out = run(a.steps, a.obstacles, a.seed, shift=a.shift)
    print({k: v for k, v in out.items() if not torch.is_tensor(v)})
