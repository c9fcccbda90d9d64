"""Test doubles: oracle-backed CPU stand-ins for `sigsvgd_amd.ops`, used ONLY by the `-m "not gpu"`
host-logic tests (the product has no CPU path; these live under tests/ on purpose)."""
import os

import numpy as np
import torch

from oracle import sigkernel_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_fixtures.npz")


def golden():
    return np.load(GOLDEN, allow_pickle=False)


def _np(t):
    return t.detach().cpu().numpy().astype(np.float64)


def gram_fwd(X, Y, inv_h, dyadic_order=0, static_kind=0, naive=False, force_generic=False, y_is_x=False):
    K = O.gram(_np(X), _np(Y), static_kind, 1.0 / inv_h, dyadic_order, naive)
    return torch.as_tensor(K, dtype=X.dtype)


def gram_fwd_bwd(X, Y, inv_h, dyadic_order=0, static_kind=0, grad_out=None, naive=False, sym=False,
                 y_is_x=False, force_generic=False):
    go = None if grad_out is None else _np(grad_out)
    K, g = O.gram_backward(_np(X), _np(Y), go, static_kind, 1.0 / inv_h, dyadic_order, naive, sym)
    return torch.as_tensor(K, dtype=X.dtype), torch.as_tensor(g, dtype=X.dtype)


def svgd_phi(K, score, grad_k, mask=None, X=None, lr=None, adagrad_state=None):
    N = K.shape[0]
    v = -((K.float() @ score.float().reshape(N, -1) - grad_k.float().reshape(N, -1)) / N)
    if mask is not None:
        v = v * torch.broadcast_to(torch.as_tensor(mask, dtype=torch.float32), score.shape).reshape(N, -1)
    if adagrad_state is not None:
        adagrad_state += (v * v).reshape(adagrad_state.shape)
        v = v / torch.sqrt(adagrad_state.reshape(N, -1) + 1e-12)
    v = v.reshape(score.shape)
    if X is not None:
        return v, X.float() - lr * v
    return v


def gram_sym_partial(X, inv_h, tile_offset, tile_stride, static_kind=0, grad_out=None, sym=False, out=None, fold=False):
    """Same ownership rule as the HIP kernels, taken from the library itself (host-only queries): unordered pairs
    {i <= j} whose row tile of i -- `ops.sym_tile_rows(T, d)` rows -- is one of `ops.owned_tiles(...)`."""
    from sigsvgd_amd import ops

    Xn = _np(X)
    N, T, d = Xn.shape
    nw = ops.sym_tile_rows(T, d)
    owned = set(ops.owned_tiles((N + nw - 1) // nw, tile_offset, tile_stride, fold))
    Kp = np.zeros((N, N))
    gp = np.zeros((N, T, d))
    for i in range(N):
        if (i // nw) not in owned:
            continue
        for j in range(i, N):
            Kij, gi = O.gram_backward(Xn[i:i + 1], Xn[j:j + 1], None, static_kind, 1.0 / inv_h, 0)
            Kp[i, j] = Kp[j, i] = Kij[0, 0]
            gp[i] += gi[0]
            if j != i:
                _, gj = O.gram_backward(Xn[j:j + 1], Xn[i:i + 1], None, static_kind, 1.0 / inv_h, 0)
                gp[j] += gj[0]
    Kt, gt = torch.as_tensor(Kp, dtype=X.dtype), torch.as_tensor(gp, dtype=torch.float64)
    if out is not None:
        out[0].copy_(Kt)
        out[1].copy_(gt)
        return out
    return Kt, gt


def patch_ops(monkeypatch):
    """Replace the HIP-backed ops by the oracle-backed doubles (CPU host-logic tests only)."""
    from sigsvgd_amd import ops

    for name, fn in [("gram_fwd", gram_fwd), ("gram_fwd_bwd", gram_fwd_bwd), ("svgd_phi", svgd_phi),
                     ("gram_sym_partial", gram_sym_partial)]:
        monkeypatch.setattr(ops, name, fn)
