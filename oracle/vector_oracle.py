"""CPU oracle (numpy, fp64) for the vector kernels and the truncated path signature -- TEST
INFRASTRUCTURE ONLY (imported by tests/, __graft_entry__.smoke() and nothing in the product path).

Pinned / unpinned:
  * gaussian / scaled_gaussian / imq / scaled_imq restate reference src/kernels/_kernels.py:64-299 and
    src/utils/math.py:69-86,116-144 and ARE pinned: tests/test_oracle_vector.py checks them against
    outputs of the reference's own classes captured in tests/golden/ref_vector_kernels.npz.
  * signature() restates the published definition of the path signature of a piecewise-linear path
    (iterated integrals; Chen's identity S(x * y) = S(x) (x) S(y), segment signature exp(increment)) with
    signatory's output convention (levels 1..depth concatenated, `basepoint=True` prepends a zero point)
    as used at reference src/kernels/_traj_kernels.py:124-125.  `signatory==1.2.6.1.9.0` (setup.py:58)
    is absent from /root/reference and cannot be installed offline: PARITY UNPINNED for the signature
    itself.  It is checked against an independent brute-force evaluation of the iterated sums and against
    closed forms (tests/test_oracle_vector.py).
"""
from __future__ import annotations

import itertools
import math

import numpy as np

from .sigkernel_oracle import bw_median


def pw_dist_sq(X, Y):
    """src/utils/math.py:69-86 (direct differences instead of the addmm expansion; clamp kept)."""
    X, Y = np.asarray(X, np.float64), np.asarray(Y, np.float64)
    diff = X[:, None, :] - Y[None, :, :]
    return np.maximum((diff * diff).sum(-1), 0.0)


def scaled_pw_dist_sq(X, Y, M):
    """src/utils/math.py:116-144: returns (diff M diff^T clamped at 0, diff @ M)."""
    X, Y, M = np.asarray(X, np.float64), np.asarray(Y, np.float64), np.asarray(M, np.float64)
    diff = X[:, None, :] - Y[None, :, :]
    diff_M = diff @ M
    return np.maximum((diff_M * diff).sum(-1), 0.0), diff_M


def _bandwidth(sq, h):
    return float(bw_median(sq)) if h is None else float(h)


def gaussian(X, Y, h=None):
    """GaussianKernel.__call__ (_kernels.py:98-111): K, d_K.sum(1), h."""
    X, Y = np.asarray(X, np.float64).reshape(len(X), -1), np.asarray(Y, np.float64).reshape(len(Y), -1)
    sq = pw_dist_sq(X, Y)
    h = _bandwidth(sq, h)
    K = np.exp(-0.5 / h**2 * sq)
    dK = -(X[:, None, :] - Y[None, :, :]) / h**2 * K[..., None]
    return K, dK.sum(1), h


def scaled_gaussian(X, Y, M=None, h=None):
    """ScaledGaussianKernel.__call__ (_kernels.py:165-186); M is symmetrised as there."""
    X, Y = np.asarray(X, np.float64).reshape(len(X), -1), np.asarray(Y, np.float64).reshape(len(Y), -1)
    M = np.eye(X.shape[1]) if M is None else 0.5 * (np.asarray(M, np.float64) + np.asarray(M, np.float64).T)
    sq, diff_M = scaled_pw_dist_sq(X, Y, M)
    h = _bandwidth(sq, h)
    K = np.exp(-0.5 / h**2 * sq)
    dK = -diff_M * K[..., None] / h**2
    return K, dK.sum(1), h


def imq(X, Y, h=None):
    """IMQKernel.__call__ (_kernels.py:217-235), including its (Y - X) orientation of d_K."""
    X, Y = np.asarray(X, np.float64).reshape(len(X), -1), np.asarray(Y, np.float64).reshape(len(Y), -1)
    sq = pw_dist_sq(X, Y)
    h = _bandwidth(sq, h)
    den = 1 + 0.5 * sq / h**2
    K = den**-0.5
    dK = -0.5 * den[..., None] ** -1.5 * ((Y[None, :, :] - X[:, None, :]) / h**2)
    return K, dK.sum(1), h


def scaled_imq(X, Y, M=None, h=None):
    """ScaledIMQKernel.__call__ (_kernels.py:270-299); M is used as given (not symmetrised)."""
    X, Y = np.asarray(X, np.float64).reshape(len(X), -1), np.asarray(Y, np.float64).reshape(len(Y), -1)
    M = np.eye(X.shape[1]) if M is None else np.asarray(M, np.float64)
    sq, diff_M = scaled_pw_dist_sq(X, Y, M)
    h = _bandwidth(sq, h)
    den = 1 + 0.5 * sq / h**2
    K = den**-0.5
    dK = -0.5 * den[..., None] ** -1.5 * (diff_M / h**2)
    return K, dK.sum(1), h


def vec_kernel_weighted_grad(sq, XM, YM, grad_out, kind, h, grad_scale):
    """grad_scale * sum_j grad_out[i,j] w(sq[i,j]) (XM_i - YM_j): what sigsvgd_vec_kernel returns."""
    sq = np.asarray(sq, np.float64)
    if kind == "gaussian":
        w = np.exp(-0.5 / h**2 * sq)
    else:
        w = (1 + 0.5 * sq / h**2) ** -1.5
    if grad_out is not None:
        w = w * np.asarray(grad_out, np.float64)
    XM, YM = np.asarray(XM, np.float64), np.asarray(YM, np.float64)
    return grad_scale * (XM * w.sum(1, keepdims=True) - w @ YM)


# ---- truncated signature ----------------------------------------------------------------------------
def signature_channels(C, depth):
    return sum(C**k for k in range(1, depth + 1))


def _tensor_exp(inc, depth):
    """exp(inc) truncated: levels 1..depth of inc^{(x)k} / k!."""
    levels, cur = [], np.ones(())
    for k in range(1, depth + 1):
        cur = np.multiply.outer(cur, inc) / k
        levels.append(cur)
    return levels


def _chen(S, E, depth):
    """(1 + S) (x) (1 + E) truncated; S, E lists of level tensors 1..depth."""
    out = []
    for k in range(1, depth + 1):
        acc = S[k - 1] + E[k - 1]
        for m in range(1, k):
            acc = acc + np.multiply.outer(S[m - 1], E[k - m - 1])
        out.append(acc)
    return out


def signature(X, depth, basepoint=True):
    """[N, L, C] -> [N, C + ... + C^depth], Chen's identity over the segments."""
    X = np.asarray(X, np.float64)
    N, L, C = X.shape
    out = np.zeros((N, signature_channels(C, depth)))
    for i in range(N):
        pts = np.concatenate([np.zeros((1, C)), X[i]], 0) if basepoint else X[i]
        S = [np.zeros((C,) * k) for k in range(1, depth + 1)]
        for t in range(1, len(pts)):
            S = _chen(S, _tensor_exp(pts[t] - pts[t - 1], depth), depth)
        out[i] = np.concatenate([s.reshape(-1) for s in S])
    return out


def signature_vjp(X, grad_sig, depth, basepoint=True):
    """d sum(grad_sig * signature(X)) / dX by reverse-mode differentiation of the SAME recursion (torch fp64 autograd over
    Chen's identity written with tensor products): the reference for the HIP adjoint `sigsvgd_signature_backward`.  What
    the reference itself runs here is signatory's backward (absent); tests pin this function with central finite
    differences of `signature` above."""
    import torch

    x = torch.tensor(np.asarray(X, np.float64), requires_grad=True)
    N, L, C = x.shape
    pts = torch.cat([torch.zeros(N, 1, C, dtype=torch.float64), x], 1) if basepoint else x
    inc = pts[:, 1:] - pts[:, :-1]
    S = [torch.zeros(N, C**k, dtype=torch.float64) for k in range(1, depth + 1)]
    for t in range(inc.shape[1]):
        D = inc[:, t]
        E = [D]
        for m in range(2, depth + 1):
            E.append((E[-1][:, :, None] * D[:, None, :]).reshape(N, -1) / m)
        new = []
        for k in range(1, depth + 1):
            acc = S[k - 1] + E[k - 1]
            for j in range(1, k):
                acc = acc + (S[j - 1][:, :, None] * E[k - j - 1][:, None, :]).reshape(N, -1)
            new.append(acc)
        S = new
    sig = torch.cat(S, 1)
    (g,) = torch.autograd.grad((sig * torch.as_tensor(np.asarray(grad_sig, np.float64))).sum(), x)
    return sig.detach().numpy(), g.numpy()


def signature_bruteforce(x, depth, basepoint=True):
    """Independent check for ONE small path [L, C]: level k as the iterated integral over the simplex
    of the piecewise-linear path = sum over non-decreasing segment tuples t_1 <= ... <= t_k of
    prod(increments) / prod(multiplicity!)  (k points on the same straight segment contribute 1/k!)."""
    x = np.asarray(x, np.float64)
    pts = np.concatenate([np.zeros((1, x.shape[1])), x], 0) if basepoint else x
    inc = np.diff(pts, axis=0)
    n, C = inc.shape
    levels = []
    for k in range(1, depth + 1):
        lvl = np.zeros((C,) * k)
        for segs in itertools.combinations_with_replacement(range(n), k):
            wgt = 1.0
            for _, grp in itertools.groupby(segs):
                wgt /= math.factorial(len(list(grp)))
            term = np.ones(())
            for s in segs:
                term = np.multiply.outer(term, inc[s])
            lvl += wgt * term
        levels.append(lvl.reshape(-1))
    return np.concatenate(levels)


def path_sig_kernel(X, Y, depth=3, h=None):
    """PathSigKernel.__call__ with the default GaussianKernel static kernel (_traj_kernels.py:119-144):
    (K, d_K.sum(1) w.r.t. the SIGNATURE features, h)."""
    return gaussian(signature(X, depth, True), signature(Y, depth, True), h)
