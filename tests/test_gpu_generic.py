"""GPU parity: generic HIP kernel vs the fp64 oracle (through the C ABI via sigsvgd_amd.ops)."""
import numpy as np
import pytest
import torch

from oracle import c_oracle as C
from oracle import sigkernel_oracle as O

pytestmark = pytest.mark.gpu

TOL = 1e-5  # north_star: within 1e-5 relative fp32


def _paths(A, T, d, seed, scale=0.3):
    rng = np.random.default_rng(seed)
    return np.cumsum(scale * rng.standard_normal((A, T, d)), axis=1).astype(np.float32)


def _rel(a, b):
    return float(np.abs(np.asarray(a, np.float64) - b).max() / np.abs(b).max())


def _relK(a, b):
    """K parity as north_star states it: max over entries of |K - K_ref| / |K_ref| (K > 0 always)"""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    # (round 4: plain relative error per entry -- rounds 2-3 floored the denominator at 0.1; the 1e-6 only keeps an exact zero
    #  out of it.  Pairs whose K is small against their grid are solved by the exact fp64 pass now: DESIGN.md section 3)
    return float((np.abs(a - b) / np.maximum(np.abs(b), 1e-6)).max())


CASES = [
    # A, B, T, d, n, kind, naive
    (5, 4, 7, 3, 0, 0, False),
    (5, 4, 7, 3, 2, 0, False),
    (3, 6, 5, 2, 5, 0, False),   # reference obstacle-field shape: T=5, depth=5
    (4, 4, 3, 7, 6, 0, False),   # reference robot shape: T=3 knots, depth=6, d=7
    (6, 5, 20, 2, 2, 0, False),  # C1: T=20, d=2, depth 2
    (3, 3, 30, 2, 3, 0, False),  # particle-maze: T=30, dyadic 3 (P=232, 4 bands)
    (4, 5, 9, 3, 1, 1, False),   # linear static kernel
    (4, 5, 9, 3, 1, 0, True),    # naive solver
    (2, 3, 70, 3, 0, 0, False),  # T > 64 (two bands at n=0)
    (3, 2, 33, 17, 0, 0, False), # d > 16
    (3, 2, 128, 14, 0, 0, False), # C5 path shape (T=128, d=14): compact-LDS mode of the generic kernel
    (2, 3, 100, 3, 0, 0, False),  # long path, three bands
]


@pytest.mark.parametrize("A,B,T,d,n,kind,naive", CASES)
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_generic_fwd_bwd(gpu, A, B, T, d, n, kind, naive, dtype):
    from sigsvgd_amd import ops

    X = _paths(A, T, d, 1)
    Y = _paths(B, T, d, 2)
    h = 1.7
    rng = np.random.default_rng(3)
    go = rng.standard_normal((A, B))
    Kref, gref = O.gram_backward(X, Y, go, kind, h, n, naive)
    Xg = torch.as_tensor(X, device=gpu).to(dtype)
    Yg = torch.as_tensor(Y, device=gpu).to(dtype)
    gog = torch.as_tensor(go, device=gpu).to(dtype)
    K1 = ops.gram_fwd(Xg, Yg, 1.0 / h, n, kind, naive=naive, force_generic=True)
    K2, g2 = ops.gram_fwd_bwd(Xg, Yg, 1.0 / h, n, kind, grad_out=gog, naive=naive, force_generic=True)
    torch.cuda.synchronize()
    assert _relK(K1.cpu().numpy(), Kref) < TOL
    assert _relK(K2.cpu().numpy(), Kref) < TOL
    # grad_out is rounded to the I/O dtype on the way in
    gref_io = O.gram_backward(X, Y, gog.cpu().numpy().astype(np.float64), kind, h, n, naive)[1]
    assert _rel(g2.cpu().numpy(), gref_io) < TOL
    # ones path (NULL grad_out)
    K3, g3 = ops.gram_fwd_bwd(Xg, Yg, 1.0 / h, n, kind, naive=naive, force_generic=True)
    assert _rel(g3.cpu().numpy(), O.gram_backward(X, Y, None, kind, h, n, naive)[1]) < TOL


def test_phi(gpu):
    from sigsvgd_amd import ops

    rng = np.random.default_rng(0)
    for N, D in [(16, 40), (100, 20), (128, 224), (257, 65)]:
        K = rng.standard_normal((N, N)).astype(np.float32)
        s = rng.standard_normal((N, D)).astype(np.float32)
        gk = rng.standard_normal((N, D)).astype(np.float32)
        m = (rng.random((N, D)) > 0.3).astype(np.float32)
        X = rng.standard_normal((N, D)).astype(np.float32)
        vref = O.svgd_velocity(K, s, gk, m)
        v, Xn = ops.svgd_phi(*(torch.as_tensor(t, device=gpu) for t in (K, s, gk, m)), X=torch.as_tensor(X, device=gpu), lr=0.1)
        assert _rel(v.cpu().numpy(), vref) < TOL
        assert _rel(Xn.cpu().numpy(), X - 0.1 * vref) < TOL
        v2 = ops.svgd_phi(*(torch.as_tensor(t, device=gpu) for t in (K, s, gk)))
        assert _rel(v2.cpu().numpy(), O.svgd_velocity(K, s, gk)) < TOL


@pytest.mark.parametrize("N,T,d,n,kind", [(70, 10, 2, 4, 0), (66, 20, 2, 2, 0), (64, 5, 3, 5, 0), (65, 30, 2, 2, 0), (67, 12, 3, 1, 1),
                                          (64, 128, 14, 0, 0), (9, 10, 2, 3, 0)])
@pytest.mark.parametrize("weights", ["ones", "random", "sym"])
def test_generic_symmetric_solve(gpu, N, T, d, n, kind, weights):
    """Y is X on the coverage kernel: pairs j >= i only, K mirrored, column-side gradient through the fp64
    accumulation buffer -- must equal the ordered-pair result of the oracle (also with asymmetric weights)."""
    from sigsvgd_amd import ops

    rng = np.random.default_rng(N * 100 + T)
    X = np.cumsum(0.1 * rng.standard_normal((N, T, d)), axis=1).astype(np.float32)
    h = 1.3
    go = None if weights == "ones" else rng.uniform(0.5, 1.5, (N, N)).astype(np.float32)
    sym = weights == "sym"
    w = None if go is None else (go + go.T if sym else go).astype(np.float64)
    if go is None and sym:
        w = np.full((N, N), 2.0)
    Kref, gref = C.gram_fwd_bwd(X, X, h, n, kind=kind, grad_out=w)
    Xg = torch.as_tensor(X, device=gpu)
    gog = None if go is None else torch.as_tensor(go, device=gpu)
    K, g = ops.gram_fwd_bwd(Xg, Xg, 1.0 / h, n, static_kind=kind, grad_out=gog, sym=sym, y_is_x=True, force_generic=True)
    assert torch.equal(K, K.T)
    assert _relK(K.cpu().numpy(), Kref) < TOL and _rel(g.cpu().numpy(), gref) < TOL
    Kf = ops.gram_fwd(Xg, Xg, 1.0 / h, n, static_kind=kind, force_generic=True, y_is_x=True)
    assert _relK(Kf.cpu().numpy(), Kref) < TOL


@pytest.mark.parametrize("T,d,n,scale,h", [(64, 1, 0, 0.5, 1.0), (100, 1, 0, 0.1, 0.1), (100, 1, 0, 0.2, 0.1), (33, 1, 2, 0.2, 0.1),
                                           (20, 1, 2, 0.2, 0.1), (64, 2, 0, 0.2, 0.1)])
def test_coverage_kernel_is_fp64_end_to_end(gpu, T, d, n, scale, h):
    """Rough paths in one channel: the discrete solution oscillates and K[P][P] is ill-conditioned with respect to the
    increments -- the fp32-sweep kernels reach 6e-5 there, and so did this kernel while it stored its
    increment table in fp32 (1.9e-5 at T = 100).  With the table in fp64 (`force_generic=True`: whenever it fits 160 KB)
    every entry is the fp64 reference's up to the fp32 store of K, and the gradient with it."""
    from sigsvgd_amd import ops

    rng = np.random.default_rng(3)
    X = np.cumsum(scale * rng.standard_normal((9, T, d)), axis=1).astype(np.float32)
    Y = np.cumsum(scale * rng.standard_normal((11, T, d)), axis=1).astype(np.float32)
    Kref, gref = C.gram_fwd_bwd(X, Y, h, n)
    Xg, Yg = torch.as_tensor(X, device=gpu), torch.as_tensor(Y, device=gpu)
    K, g = ops.gram_fwd_bwd(Xg, Yg, 1.0 / h, n, force_generic=True)
    Kf = ops.gram_fwd(Xg, Yg, 1.0 / h, n, force_generic=True)
    assert _relK(K.cpu().numpy(), Kref) < 2e-7
    assert _relK(Kf.cpu().numpy(), Kref) < 2e-7
    assert _rel(g.cpu().numpy(), gref) < 1e-6
    K64, g64 = ops.gram_fwd_bwd(Xg.double(), Yg.double(), 1.0 / h, n, force_generic=True)
    assert _relK(K64.cpu().numpy(), Kref) < 1e-10
