"""Natural cubic splines through trajectory knots, evaluated for a whole particle batch on the device --
the step in front of the signature-kernel path in the reference's planning scripts
(`create_spline_trajectory`, examples/script_planning_obstacle_field.py:18-23, script_planning_robot.py:117-121),
which get it from the third-party `torchcubicspline` (absent from the reference tree and this image).

Same call shape as that package for the two functions the reference uses:

    coeffs = natural_cubic_spline_coeffs(t_knots, knots)      # knots [..., K, d]
    traj = NaturalCubicSpline(coeffs).evaluate(t)              # [..., len(t), d]

For fixed knot times the spline is a LINEAR map of the knot values, so evaluation is one small GEMM with a
[len(t), K] basis matrix (built once per (t_knots, t) pair in fp64 and cached); it runs on whatever device the
knots live on and is differentiable through torch autograd, which is all the cost functions in front of the
path need.  The natural cubic spline through given knots is unique, so parity with torchcubicspline is a
matter of rounding; the tests pin it against scipy.interpolate.CubicSpline(bc_type="natural").
"""
from __future__ import annotations

from typing import Tuple

import torch


def _second_derivative_operator(t_knots: torch.Tensor) -> torch.Tensor:
    """[K, K] matrix S with M = S @ y: the knot second derivatives of the natural spline (M_0 = M_{K-1} = 0)."""
    tk = t_knots.double().cpu()
    K = tk.numel()
    if K < 2:
        raise ValueError("a spline needs at least two knots")
    if not bool((tk[1:] > tk[:-1]).all()):
        raise ValueError("knot times must be strictly increasing")
    h = tk[1:] - tk[:-1]
    A = torch.zeros(K, K, dtype=torch.float64)
    R = torch.zeros(K, K, dtype=torch.float64)
    A[0, 0] = A[K - 1, K - 1] = 1.0
    for i in range(1, K - 1):
        A[i, i - 1], A[i, i], A[i, i + 1] = h[i - 1], 2.0 * (h[i - 1] + h[i]), h[i]
        R[i, i - 1], R[i, i], R[i, i + 1] = 6.0 / h[i - 1], -6.0 / h[i - 1] - 6.0 / h[i], 6.0 / h[i]
    return torch.linalg.solve(A, R)


def spline_basis(t_knots: torch.Tensor, t: torch.Tensor, order: int = 0) -> torch.Tensor:
    """[len(t), K] matrix B (fp64, CPU) with spline^{(order)}(t) = B @ y for knot values y; order 0, 1 or 2.
    Outside [t_0, t_{K-1}] the first / last polynomial piece is extended, as torchcubicspline does."""
    tk = t_knots.double().cpu()
    te = t.double().cpu().reshape(-1)
    K = tk.numel()
    S = _second_derivative_operator(tk)
    idx = torch.clamp(torch.bucketize(te, tk, right=True) - 1, 0, K - 2)
    h = (tk[1:] - tk[:-1])[idx]
    a = (tk[idx + 1] - te) / h  # weight of the left knot
    b = (te - tk[idx]) / h      # weight of the right knot
    E = torch.zeros(te.numel(), K, dtype=torch.float64)
    rows = torch.arange(te.numel())
    if order == 0:
        wl, wr = a, b
        ml, mr = (a**3 - a) * h * h / 6.0, (b**3 - b) * h * h / 6.0
    elif order == 1:
        wl, wr = -1.0 / h, 1.0 / h
        ml, mr = -(3.0 * a * a - 1.0) * h / 6.0, (3.0 * b * b - 1.0) * h / 6.0
    elif order == 2:
        wl = wr = torch.zeros_like(a)
        ml, mr = a, b
    else:
        raise ValueError("order must be 0, 1 or 2")
    E[rows, idx] += wl
    E[rows, idx + 1] += wr
    return E + ml[:, None] * S[idx] + mr[:, None] * S[idx + 1]


def natural_cubic_spline_coeffs(t_knots: torch.Tensor, knots: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Mirror of torchcubicspline.natural_cubic_spline_coeffs(t, x): x [..., K, channels] on any device.
    The returned object is opaque to callers (they hand it to NaturalCubicSpline)."""
    if knots.shape[-2] != t_knots.numel():
        raise ValueError(f"knots have {knots.shape[-2]} points but t has {t_knots.numel()}")
    return (t_knots, knots)


class NaturalCubicSpline:
    """evaluate(t) / derivative(t, order) of the natural cubic spline through `knots` at times `t_knots`."""

    def __init__(self, coeffs):
        self.t_knots, self.knots = coeffs
        self._cache = {}

    def _basis(self, t: torch.Tensor, order: int) -> torch.Tensor:
        t = torch.as_tensor(t)
        key = (order, tuple(t.shape), hash(t.detach().double().cpu().numpy().tobytes()))
        B = self._cache.get(key)
        if B is None:
            B = spline_basis(self.t_knots, t, order).to(device=self.knots.device, dtype=self.knots.dtype)
            self._cache = {key: B}
        return B

    def evaluate(self, t: torch.Tensor) -> torch.Tensor:
        """[..., K, d] knots -> [..., len(t), d] (a scalar t gives [..., d])."""
        t = torch.as_tensor(t)
        out = self._basis(t, 0) @ self.knots
        return out.squeeze(-2) if t.dim() == 0 else out

    def derivative(self, t: torch.Tensor, order: int = 1) -> torch.Tensor:
        t = torch.as_tensor(t)
        out = self._basis(t, order) @ self.knots
        return out.squeeze(-2) if t.dim() == 0 else out


def create_spline_trajectory(knots: torch.Tensor, timesteps: int = 100) -> torch.Tensor:
    """The reference's helper (script_planning_obstacle_field.py:18-23): uniform knot times on [0, 1],
    `timesteps` uniform samples; [batch, K, d] -> [batch, timesteps, d] on the knots' device."""
    t = torch.linspace(0, 1, timesteps)
    t_knots = torch.linspace(0, 1, knots.shape[-2])
    return NaturalCubicSpline(natural_cubic_spline_coeffs(t_knots, knots)).evaluate(t)
