"""`sigkernel`-shaped CPU objects backed by the numpy oracle (test infrastructure only).

Gives the oracle the same Python surface as the absent third-party package
(`SigKernel(static_kernel, dyadic_order).compute_Gram(X, Y, sym=False)`, `RBFKernel(sigma)`,
`LinearKernel()`), as an autograd node whose backward returns a gradient for X only -- the contract
the reference's callers rely on (/root/reference/src/inference/score.py:68-69,
/root/reference/src/inference/trajectory_svgd.py:55-65).  tests/golden/make_golden.py registers this
module under the name `sigkernel` so that the REFERENCE's own ScoreEstimator / TrajectorySVGD /
SignatureKernel code can run on top of it and its outputs can be captured as fixtures.
"""
from __future__ import annotations

import numpy as np
import torch

from . import sigkernel_oracle as O


class LinearKernel:
    kind = O.LINEAR

    def bandwidth(self, X, Y):
        return 1.0

    def batch_kernel(self, X, Y):
        return torch.bmm(X, Y.permute(0, 2, 1))

    def Gram_matrix(self, X, Y):
        return torch.einsum("ipk,jqk->ijpq", X, Y)


class RBFKernel:
    kind = O.RBF

    def __init__(self, sigma):
        self.sigma = sigma

    def bandwidth(self, X, Y):
        return float(self.sigma)

    def batch_kernel(self, X, Y):
        return torch.as_tensor(O.static_batch(X.detach().numpy(), Y.detach().numpy(), O.RBF, self.sigma))

    def Gram_matrix(self, X, Y):
        return torch.as_tensor(O.static_gram(X.detach().numpy(), Y.detach().numpy(), O.RBF, self.sigma))


def _resolve_static(static_kernel, X, Y):
    """(kind, h) for a static kernel object: ours, or the reference's BatchGaussianKernel
    (get_bandwidth evaluated on the full [A,B,T,T] distance tensor, _traj_kernels.py:191-194)."""
    if hasattr(static_kernel, "kind"):
        return static_kernel.kind, static_kernel.bandwidth(X, Y)
    if hasattr(static_kernel, "get_bandwidth"):
        dist = torch.as_tensor(O.pairwise_sqdist(X.detach().numpy(), Y.detach().numpy()))
        return O.RBF, float(static_kernel.get_bandwidth(dist))
    raise TypeError(f"unsupported static kernel {type(static_kernel)}")


class _OracleGram(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X, Y, kind, h, n, naive, sym):
        K = O.gram(X.detach().numpy(), Y.detach().numpy(), kind, h, n, naive)
        ctx.save_for_backward(X.detach(), Y.detach())
        ctx.cfg = (kind, h, n, naive, sym)
        return torch.as_tensor(K, dtype=X.dtype)

    @staticmethod
    def backward(ctx, grad_output):
        X, Y = ctx.saved_tensors
        kind, h, n, naive, sym = ctx.cfg
        _, gX = O.gram_backward(X.numpy(), Y.numpy(), grad_output.detach().numpy(), kind, h, n, naive, sym)
        return torch.as_tensor(gX, dtype=X.dtype), None, None, None, None, None, None


class SigKernel:
    def __init__(self, static_kernel, dyadic_order, _naive_solver=False):
        self.static_kernel = static_kernel
        self.dyadic_order = dyadic_order
        self._naive_solver = _naive_solver

    def compute_Gram(self, X, Y, sym=False):
        kind, h = _resolve_static(self.static_kernel, X, Y)
        return _OracleGram.apply(X, Y, kind, h, self.dyadic_order, self._naive_solver, sym)
