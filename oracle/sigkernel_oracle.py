"""CPU ORACLE (test infrastructure, NOT product code) for the sig-kernel SVGD hot path.

PARITY STATUS: *parity unpinned* at the per-entry numeric level.  The PDE arithmetic of the
reference lives in the third-party package `sigkernel`, pinned in /root/reference/setup.py:71 as
`crispitagorico/sigkernel@3b2373982e12b3d499a80228311a04debcc1bea1`, which is NOT present in
/root/reference nor installed, and no reference test stores a usable Gram/gradient value
(SURVEY.md §4, §8c).  This file restates the published algorithm (Salvi et al., "The Signature
Kernel is the solution of a Goursat PDE"; second-order explicit stencil; variation-of-parameters
gradient `K_fwd * K_rev`) anchored on the reference's own call sites:

  * static kernel  ............ /root/reference/src/kernels/_traj_kernels.py:176-195 (`exp(-dist/h)`)
  * Gram call  ................ /root/reference/src/kernels/_traj_kernels.py:203-206
                                 /root/reference/src/inference/trajectory_svgd.py:60-65
  * grad_k = d(sum K)/dX ...... /root/reference/src/inference/score.py:68-69
  * SVGD velocity / step ...... /root/reference/src/inference/svgd.py:82-83, 106-115
  * bandwidth heuristic ....... /root/reference/src/utils/math.py:28-34

Everything in the tree that the path touches (static kernel, SVGD.step, schedulers, bw_median) IS
pinned: tests/golden/*.npz holds outputs of the reference's own Python for those pieces (generated
by tests/golden/make_golden.py in the build container) and tests/test_oracle.py replays them.
The PDE part is pinned only by known-answer tests (closed forms, invariances, adjoint identity,
finite differences) in tests/test_oracle.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

All arithmetic is float64 numpy.  Shapes: X [A,T,d], Y [B,T,d]; n = dyadic order, r = 2**n,
P = r*(T-1).
"""
from __future__ import annotations

import numpy as np

RBF = 0
LINEAR = 1


# --------------------------------------------------------------------------------------------
# static kernel  (reference: src/kernels/_traj_kernels.py:176-195; sigkernel RBFKernel/LinearKernel)
# --------------------------------------------------------------------------------------------
def pairwise_sqdist(X: np.ndarray, Y: np.ndarray) -> np.ndarray:
    """dist[i,j,p,q] = |X_ip|^2 + |Y_jq|^2 - 2<X_ip,Y_jq>; NOT clamped (as the reference)."""
    X = np.asarray(X, dtype=np.float64)
    Y = np.asarray(Y, dtype=np.float64)
    Xs = (X**2).sum(-1)
    Ys = (Y**2).sum(-1)
    dist = -2.0 * np.einsum("ipk,jqk->ijpq", X, Y)
    dist += Xs[:, None, :, None] + Ys[None, :, None, :]
    return dist


def static_gram(X, Y, kind: int = RBF, h: float = 1.0) -> np.ndarray:
    """G[i,j,p,q] = k(X_ip, Y_jq).  RBF: exp(-dist/h)  (note: /h, not /(2h^2))."""
    X = np.asarray(X, dtype=np.float64)
    Y = np.asarray(Y, dtype=np.float64)
    if kind == LINEAR:
        return np.einsum("ipk,jqk->ijpq", X, Y)
    return np.exp(-pairwise_sqdist(X, Y) / float(h))


def static_batch(X, Y, kind: int = RBF, h: float = 1.0) -> np.ndarray:
    """Paired variant k(X^i_s, Y^i_t) (reference `batch_kernel`, _traj_kernels.py:156-174)."""
    X = np.asarray(X, dtype=np.float64)
    Y = np.asarray(Y, dtype=np.float64)
    if kind == LINEAR:
        return np.einsum("ipk,iqk->ipq", X, Y)
    Xs = (X**2).sum(-1)
    Ys = (Y**2).sum(-1)
    dist = -2.0 * np.einsum("ipk,iqk->ipq", X, Y) + Xs[:, :, None] + Ys[:, None, :]
    return np.exp(-dist / float(h))


def bw_median(sq_dists: np.ndarray, bw_scale: float = 1.0, tol: float = 1e-8) -> float:
    """reference src/utils/math.py:28-34 (torch.median = LOWER median of the flattened tensor)."""
    flat = np.sort(np.asarray(sq_dists, dtype=np.float64).ravel())
    med = flat[(flat.size - 1) // 2]
    # the reference takes the log of a float32 0-dim tensor (torch.tensor(rows + 1.0).log()), so the
    # divisor carries float32 rounding; reproduce it
    h = med / np.float64(np.log(np.float32(sq_dists.shape[0] + 1.0), dtype=np.float32))
    h = bw_scale * np.sqrt(h)
    return float(max(h, tol))


# --------------------------------------------------------------------------------------------
# increments + dyadic refinement + Goursat sweep   [RECALLED: sigkernel _SigKernelGram.forward]
# --------------------------------------------------------------------------------------------
def increments(G: np.ndarray) -> np.ndarray:
    """D[...,a,b] = G[a+1,b+1] + G[a,b] - G[a+1,b] - G[a,b+1]."""
    return G[..., 1:, 1:] + G[..., :-1, :-1] - G[..., 1:, :-1] - G[..., :-1, 1:]


def refine(D: np.ndarray, n: int) -> np.ndarray:
    """g[...,p,q] = D[...,p//r,q//r] / r^2."""
    r = 2**n
    if r == 1:
        return D
    g = np.repeat(np.repeat(D, r, axis=-2), r, axis=-1)
    return g / float(r * r)


def pde_sweep(g: np.ndarray, naive: bool = False) -> np.ndarray:
    """Solve the Goursat PDE on increments g [...,P,Q] -> K [...,P+1,Q+1], K[0,:]=K[:,0]=1.

    default:  K[p+1,q+1] = (K[p+1,q] + K[p,q+1])*(1 + g/2 + g^2/12) - K[p,q]*(1 - g^2/12)
    naive:    K[p+1,q+1] =  K[p+1,q] + K[p,q+1] + K[p,q]*(g - 1)
    Vectorised over anti-diagonals (cells on one anti-diagonal are independent).
    """
    g = np.asarray(g, dtype=np.float64)
    P, Q = g.shape[-2], g.shape[-1]
    K = np.ones(g.shape[:-2] + (P + 1, Q + 1), dtype=np.float64)
    for s in range(P + Q - 1):
        p = np.arange(max(0, s - Q + 1), min(P, s + 1))
        q = s - p
        gg = g[..., p, q]
        k10 = K[..., p + 1, q]
        k01 = K[..., p, q + 1]
        k00 = K[..., p, q]
        if naive:
            K[..., p + 1, q + 1] = k10 + k01 + k00 * (gg - 1.0)
        else:
            g2 = gg * gg / 12.0
            K[..., p + 1, q + 1] = (k10 + k01) * (1.0 + 0.5 * gg + g2) - k00 * (1.0 - g2)
    return K


def pde_sweep_scalar(g: np.ndarray, naive: bool = False) -> np.ndarray:
    """Plain double loop for ONE pair (the literal upstream cython loop order); small cases only."""
    P, Q = g.shape
    K = np.ones((P + 1, Q + 1))
    for p in range(P):
        for q in range(Q):
            x = g[p, q]
            if naive:
                K[p + 1, q + 1] = K[p + 1, q] + K[p, q + 1] + K[p, q] * (x - 1.0)
            else:
                K[p + 1, q + 1] = (K[p + 1, q] + K[p, q + 1]) * (1.0 + 0.5 * x + x * x / 12.0) - K[
                    p, q
                ] * (1.0 - x * x / 12.0)
    return K


def gram_forward_full(X, Y, kind=RBF, h=1.0, n=0, naive=False):
    """Returns (K_full [A,B,P+1,P+1], g [A,B,P,P], G [A,B,T,T])."""
    G = static_gram(X, Y, kind, h)
    g = refine(increments(G), n)
    return pde_sweep(g, naive), g, G


def gram(X, Y, kind=RBF, h=1.0, n=0, naive=False) -> np.ndarray:
    """Signature-kernel Gram matrix K[i,j] (what `compute_Gram` returns)."""
    return gram_forward_full(X, Y, kind, h, n, naive)[0][..., -1, -1]


# --------------------------------------------------------------------------------------------
# backward  [RECALLED: sigkernel _SigKernelGram.backward]
# --------------------------------------------------------------------------------------------
def gg_matrix(K_full: np.ndarray, g: np.ndarray, naive=False) -> np.ndarray:
    """GG[p,q] = K_fwd[p,q] * K_rev[p+1,q+1]; K_rev = flip(sweep(flip(g)))."""
    K_rev = pde_sweep(g[..., ::-1, ::-1], naive)[..., ::-1, ::-1]
    return K_full[..., :-1, :-1] * K_rev[..., 1:, 1:]


def static_grad_x(X, Y, G, kind=RBF, h=1.0) -> np.ndarray:
    """V[i,j,m,n,c] = d k(X_im, Y_jn) / d X_im[c]   (analytic)."""
    X = np.asarray(X, dtype=np.float64)
    Y = np.asarray(Y, dtype=np.float64)
    if kind == LINEAR:
        return np.broadcast_to(Y[None, :, None, :, :], G.shape + (X.shape[-1],)).copy()
    diff = X[:, None, :, None, :] - Y[None, :, None, :, :]
    return (-2.0 / float(h)) * diff * G[..., None]


def gram_backward(X, Y, grad_out=None, kind=RBF, h=1.0, n=0, naive=False, sym=False):
    """grad_X [A,T,d] of sum(grad_out * K) w.r.t. the FIRST argument only, the reference's way:
    GG is used as dLoss/dg whatever the stencil (it is the exact adjoint only for the naive
    stencil), then chained exactly through g -> D -> G -> X.
    Returns (K [A,B], grad_X [A,T,d])."""
    X = np.asarray(X, dtype=np.float64)
    Y = np.asarray(Y, dtype=np.float64)
    A, T, d = X.shape
    B = Y.shape[0]
    K_full, g, G = gram_forward_full(X, Y, kind, h, n, naive)
    GG = gg_matrix(K_full, g, naive)
    r = 2**n
    # S = dLoss/dD: block-sum / r^2
    S = GG.reshape(A, B, T - 1, r, Y.shape[1] - 1, r).sum(axis=(3, 5)) / float(r * r)
    # R = dLoss/dG: scatter of the 4-corner stencil
    R = np.zeros((A, B, T, Y.shape[1]))
    R[:, :, 1:, 1:] += S
    R[:, :, :-1, :-1] += S
    R[:, :, 1:, :-1] -= S
    R[:, :, :-1, 1:] -= S
    V = static_grad_x(X, Y, G, kind, h)  # [A,B,T,T,d]
    pts = np.einsum("ijmn,ijmnc->ijmc", R, V)  # per-pair gradient w.r.t. X_i
    if grad_out is None:
        grad_out = np.ones((A, B))
    grad_out = np.asarray(grad_out, dtype=np.float64)
    if sym:
        w = grad_out + grad_out.T
    else:
        w = grad_out
    grad_X = np.einsum("ij,ijmc->imc", w, pts)
    return K_full[..., -1, -1], grad_X


def gram_backward_fd_literal(X, Y, grad_out=None, h=1.0, n=0, naive=False, fd=1e-9):
    """[RECALLED] literal restatement of the upstream backward dataflow (RBF only): static-kernel
    derivative by forward finite differences (step 1e-9, fp64) on the (A, T*d, d) perturbed paths,
    Diff_1 / Diff_2 / grad_points assembly.  Used only to show that `gram_backward` (closed form)
    computes the same thing (up to FD noise ~1e-7).  Small cases only (O(A*B*P*P*d) memory)."""
    X = np.asarray(X, dtype=np.float64)
    Y = np.asarray(Y, dtype=np.float64)
    A, M, D = X.shape
    B, N, _ = Y.shape
    r = 2**n
    K_full, g, G = gram_forward_full(X, Y, RBF, h, n, naive)
    GG = gg_matrix(K_full, g, naive)  # [A,B,P,Q]

    def tile2(Z):  # tile both time axes of a [A,B,M-1,N-1,D] tensor, dividing by r each time
        Z = np.repeat(Z, r, axis=2) / float(r)
        Z = np.repeat(Z, r, axis=3) / float(r)
        return Z

    Xh = X[:, :, :, None] + fd * np.eye(D)[None, None, :]  # [A,M,D(coord),D(pert)] -> perturb
    Xh = np.transpose(Xh, (0, 1, 3, 2)).reshape(A, M * D, D)
    G_h = static_gram(Xh, Y, RBF, h).reshape(A, B, M, D, N)
    G_h = np.transpose(G_h, (0, 1, 2, 4, 3))  # [A,B,M,N,D]
    Gs = G[..., None]
    Diff_1 = G_h[:, :, 1:, 1:] - G_h[:, :, 1:, :-1] - Gs[:, :, 1:, 1:] + Gs[:, :, 1:, :-1]
    Diff_2 = Diff_1 - G_h[:, :, :-1, 1:] + G_h[:, :, :-1, :-1] + Gs[:, :, :-1, 1:] - Gs[:, :, :-1, :-1]
    Diff_1 = tile2(Diff_1)
    Diff_2 = tile2(Diff_2)
    grad_1 = (GG[..., None] * Diff_1 / fd).sum(axis=3).reshape(A, B, M - 1, r, D).sum(axis=3)
    grad_2 = (GG[..., None] * Diff_2 / fd).sum(axis=3).reshape(A, B, M - 1, r, D).sum(axis=3)
    grad_prev = grad_1[:, :, :-1] + grad_2[:, :, 1:]
    grad_incr = grad_prev - grad_1[:, :, 1:]
    grad_points = np.concatenate(
        [(grad_2[:, :, 0] - grad_1[:, :, 0])[:, :, None], grad_incr, grad_1[:, :, -1][:, :, None]],
        axis=2,
    )
    if grad_out is None:
        grad_out = np.ones((A, B))
    return K_full[..., -1, -1], np.einsum("ij,ijmc->imc", np.asarray(grad_out, np.float64), grad_points)


# --------------------------------------------------------------------------------------------
# SVGD layer  (reference: src/inference/svgd.py:82-83, 106-115; trajectory_svgd.py:80-84)
# --------------------------------------------------------------------------------------------
def svgd_velocity(K, score, grad_k, mask=None) -> np.ndarray:
    """v = -((K @ score - grad_k) / N)   (svgd.py:82-83); optional gradient mask."""
    K = np.asarray(K, np.float64)
    N = K.shape[0]
    s = np.asarray(score, np.float64).reshape(N, -1)
    gk = np.asarray(grad_k, np.float64).reshape(N, -1)
    v = -((K @ s - gk) / N)
    if mask is not None:
        v = v * np.asarray(mask, np.float64).reshape(-1, v.shape[1]) if np.ndim(mask) else v * mask
    return v.reshape(np.shape(score))


def svgd_step_manual(X, score, K, grad_k, lr, inertia=None, mask=None):
    """optimizer=None branch of SVGD.step (svgd.py:108-115): X - lr*v, optional home-made Adagrad.
    Returns (X_new, v_used, inertia_new)."""
    v = svgd_velocity(K, score, grad_k, mask)
    if inertia is not None:
        inertia = inertia + v**2
        v = v / np.sqrt(inertia + 1e-12)
    return np.asarray(X, np.float64) - lr * v, v, inertia


def svgd_iteration(X, score, h=1.0, n=0, lr=1e-3, kind=RBF, naive=False):
    """One full hot-path iteration as the benchmark defines it:
    K = Gram(X,X); grad_k = d(sum K)/dX (first slot); phi = (K@score - grad_k)/N; X += lr*phi.
    Returns dict(K, grad_k, phi, X_new)."""
    K, gk = gram_backward(X, X, None, kind, h, n, naive)
    N = X.shape[0]
    phi = (K @ np.asarray(score, np.float64).reshape(N, -1) - gk.reshape(N, -1)) / N
    phi = phi.reshape(X.shape)
    return {"K": K, "grad_k": gk, "phi": phi, "X_new": np.asarray(X, np.float64) + lr * phi}


def synthetic_inputs(N, T, d, seed_x=0, seed_s=1):
    """Benchmark inputs of SURVEY.md §8d, generated with torch's CPU generator so that every box sees
    bit-identical particles: X = cumsum(0.05*randn) (fp64 -> fp32), score = randn (fp32)."""
    import torch

    gx = torch.Generator(device="cpu").manual_seed(seed_x)
    X = torch.cumsum(0.05 * torch.randn(N, T, d, generator=gx, dtype=torch.float64), dim=1).float()
    gs = torch.Generator(device="cpu").manual_seed(seed_s)
    score = torch.randn(N, T, d, generator=gs, dtype=torch.float32)
    return X, score
