"""Trajectory kernels: `BatchGaussianKernel` (static RBF on path points) and `SignatureKernel` (signature
kernel via the Goursat PDE) -- reference src/kernels/_traj_kernels.py:147-206 -- backed by the HIP library
instead of `sigkernel`; `PathSigKernel` (static kernel on truncated signatures, :72-144) backed by the HIP
signature kernel instead of `signatory`; `TrajectoryKernel` (:14-69)."""
from __future__ import annotations

import torch

from .. import _lib
from ..sigkernel import SigKernel, gram_sqdist, inv_bandwidth_from_fn
from ._kernels import BaseKernel, GaussianKernel, _SqDist, kernel_output, scalar_function

class TrajectoryKernel(BaseKernel):
    """RBF kernel between state-space projections phi(actions) of the inputs; the gradient w.r.t. the
    actions comes from autograd through the caller's rollout graph (reference _traj_kernels.py:14-69).
    The pairwise distance is a HIP autograd node (`_SqDist`)."""

    def __init__(self, bandwidth_fn: scalar_function = None, **kwargs):
        super().__init__(bandwidth_fn, analytic_grad=False, **kwargs)

    def __call__(self, X, Y, X_actions, h: float = None, compute_grad=True, **kwargs) -> kernel_output:
        assert X.shape == Y.shape, "X and Y must have the same dimensions."
        sq_dists = _SqDist.apply(torch.atleast_2d(X), torch.atleast_2d(Y), None)
        h = self.get_bandwidth(sq_dists) if h is None else float(h)
        gamma = -0.5 / h**2
        K = (gamma * sq_dists).exp()
        if compute_grad:
            d_K = torch.autograd.grad(K.sum(), X_actions, retain_graph=True)[0]
            return K, d_K
        return K


class PathSigKernel(BaseKernel):
    """Static kernel on truncated path signatures S(X, depth) with a zero base point (reference
    _traj_kernels.py:72-144).  As in the reference, `h` and `ref_vector` are accepted but NOT forwarded:
    the static kernel uses its own bandwidth function, and the returned gradient is the static kernel's
    `d_K.sum(1)` with respect to the SIGNATURE features ([batch, C + ... + C^depth])."""

    def __init__(self, bandwidth_fn: scalar_function = None, static_kernel: BaseKernel = None, **kwargs):
        super().__init__(bandwidth_fn, analytic_grad=False, **kwargs)
        self.static_kernel = GaussianKernel() if static_kernel is None else static_kernel

    def __call__(self, X, Y, ref_vector=None, depth: int = 3, h: float = None, compute_grad=True,
                 **kwargs) -> kernel_output:
        from .. import ops

        assert X.shape == Y.shape, "X and Y must have the same dimensions."
        X, Y = torch.atleast_3d(X), torch.atleast_3d(Y)
        X_sig = ops.signature(X, depth, basepoint=True)
        Y_sig = ops.signature(Y, depth, basepoint=True)
        if compute_grad:
            K, d_K = self.static_kernel(X_sig, Y_sig, compute_grad=True)
            return K, d_K
        return self.static_kernel(X_sig, Y_sig, compute_grad=False)


class BatchGaussianKernel(BaseKernel):
    """RBF kernel on path points, k(x, y) = exp(-|x-y|^2 / h), h = bandwidth_fn(dist)."""

    static_kind = _lib.STATIC_RBF

    def __init__(self, bandwidth_fn: scalar_function = None, **kwargs):
        super().__init__(bandwidth_fn, analytic_grad=False, **kwargs)

    def __call__(self, X: torch.Tensor, Y: torch.Tensor, **kwargs) -> kernel_output:
        return self.batch_kernel(X, Y, **kwargs)

    def batch_kernel(self, X, Y, h=None):
        """X [batch, len_X, dim], Y [batch, len_Y, dim] -> k(X^i_s, Y^i_t) [batch, len_X, len_Y]."""
        A, M, N = X.shape[0], X.shape[1], Y.shape[1]
        Xs = torch.sum(X**2, dim=2)
        Ys = torch.sum(Y**2, dim=2)
        dist = -2.0 * torch.bmm(X, Y.permute(0, 2, 1))
        dist += torch.reshape(Xs, (A, M, 1)) + torch.reshape(Ys, (A, 1, N))
        h = self.get_bandwidth(dist) if h is None else float(h)
        return torch.exp(-dist / h)

    def Gram_matrix(self, X, Y, h=None):
        """X [A, len_X, dim], Y [B, len_Y, dim] -> k(X^i_s, Y^j_t) [A, B, len_X, len_Y]."""
        dist = gram_sqdist(X, Y)
        h = self.get_bandwidth(dist) if h is None else float(h)
        return torch.exp(-dist / h)

    def inv_bandwidth(self, X, Y) -> float:
        """1/h for the fused HIP path (constant bandwidth functions never form the distance tensor)."""
        return inv_bandwidth_from_fn(self.get_bandwidth, X, Y)


class SignatureKernel:
    """Signature kernel with an RBF static kernel; `depth` is the DYADIC ORDER of the PDE solver
    (not a truncation level).  Unknown keyword arguments are accepted and ignored, as in the
    reference (examples/script_planning_robot.py:391 passes `bandwidth=`)."""

    def __init__(self, bandwidth_fn: scalar_function = None, depth: int = 3, **kwargs):
        static_kernel = BatchGaussianKernel(bandwidth_fn=bandwidth_fn)
        self.kernel = SigKernel(static_kernel, dyadic_order=depth)

    def __call__(self, X, Y, **kwargs):
        # The reference upcasts to fp64 around compute_Gram and casts K back; the HIP kernels read fp32/fp64 paths
        # directly (static kernel and increments in fp64, sweeps in fp32 difference form or fp64 depending on the
        # kernel: DESIGN.md §3), so no copies are needed.
        return self.kernel.compute_Gram(X, Y)

    def gram_and_grad(self, X, grad_out=None):
        """(K, grad_k) in one fused launch == `K = self(X, X.detach()); ag(K.sum(), X)`."""
        return self.kernel.gram_and_grad(X, None, grad_out)
