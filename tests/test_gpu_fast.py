"""GPU parity: register-resident fast kernels (n=0, T<=64, RBF) vs the fp64 oracle, via the C ABI."""
import numpy as np
import pytest
import torch

from oracle import c_oracle as C
from oracle import sigkernel_oracle as O

pytestmark = pytest.mark.gpu

TOL = 1e-5  # north_star tolerance (relative; gradients relative to max-abs)


def _paths(A, T, d, seed, scale=0.05, offset=0.0):
    rng = np.random.default_rng(seed)
    return (np.cumsum(scale * rng.standard_normal((A, T, d)), axis=1) + offset).astype(np.float32)


def _rel(a, b):
    return float(np.abs(np.asarray(a, np.float64) - b).max() / np.abs(b).max())


SHAPES = [
    # A, B, T, d
    (9, 11, 64, 7),   # headline path shape, ragged row tile / column chunk
    (8, 8, 64, 7),
    (17, 5, 32, 7),   # C2 path shape
    (10, 13, 64, 3),  # C3 path shape
    (6, 7, 20, 2),
    (5, 9, 3, 7),     # shortest supported path (T=3)
    (7, 6, 33, 14),   # d > 8 variant (6-wave workgroups)
    (3, 4, 50, 1),
]


@pytest.mark.parametrize("A,B,T,d", SHAPES)
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_fast_general_xy(gpu, A, B, T, d, dtype):
    """X != Y: ordered pairs, row-side gradient only, arbitrary grad_out."""
    from sigsvgd_amd import ops

    X, Y = _paths(A, T, d, 1), _paths(B, T, d, 2)
    h = 1.3
    go = np.random.default_rng(3).standard_normal((A, B)).astype(np.float32)
    Kref, gref = C.gram_fwd_bwd(X, Y, h, 0, grad_out=go.astype(np.float64))
    Xg, Yg, gog = (torch.as_tensor(t, device=gpu).to(dtype) for t in (X, Y, go))
    K1 = ops.gram_fwd(Xg, Yg, 1.0 / h)
    K2, g2 = ops.gram_fwd_bwd(Xg, Yg, 1.0 / h, grad_out=gog)
    torch.cuda.synchronize()
    assert _rel(K1.cpu().numpy(), Kref) < TOL
    assert _rel(K2.cpu().numpy(), Kref) < TOL
    assert _rel(g2.cpu().numpy(), gref) < TOL
    # and it agrees with the generic kernel
    K3, g3 = ops.gram_fwd_bwd(Xg, Yg, 1.0 / h, grad_out=gog, force_generic=True)
    assert _rel(g2.cpu().numpy(), g3.double().cpu().numpy()) < TOL


@pytest.mark.parametrize("N,T,d", [(19, 64, 7), (8, 64, 7), (33, 32, 7), (12, 64, 3), (16, 20, 2), (13, 40, 14)])
@pytest.mark.parametrize("weights", ["ones", "random", "sym"])
def test_fast_symmetric(gpu, N, T, d, weights):
    """Y is X: unordered pairs solved once, row- and column-side gradients."""
    from sigsvgd_amd import ops

    X = _paths(N, T, d, 5)
    h = 0.9
    go = None
    sym = False
    if weights != "ones":
        go = np.random.default_rng(7).standard_normal((N, N)).astype(np.float32)
    if weights == "sym":
        sym = True
    Kref, gref = O.gram_backward(X, X, None if go is None else go.astype(np.float64), O.RBF, h, 0, False, sym)
    Xg = torch.as_tensor(X, device=gpu)
    gog = None if go is None else torch.as_tensor(go, device=gpu)
    K, g = ops.gram_fwd_bwd(Xg, Xg, 1.0 / h, grad_out=gog, sym=sym, y_is_x=True)
    torch.cuda.synchronize()
    Kn = K.cpu().numpy()
    assert _rel(Kn, Kref) < TOL
    assert np.array_equal(Kn, Kn.T)  # mirrored entries are the same solve
    assert _rel(g.cpu().numpy(), gref) < TOL


def test_fast_far_from_origin(gpu):
    """Per-pair centring: particles far from the origin keep full accuracy."""
    from sigsvgd_amd import ops

    X = _paths(12, 64, 7, 11, offset=100.0)
    Kref, gref = O.gram_backward(X, X, None, O.RBF, 1.0, 0)
    K, g = ops.gram_fwd_bwd(torch.as_tensor(X, device=gpu), torch.as_tensor(X, device=gpu), 1.0, y_is_x=True)
    assert _rel(K.cpu().numpy(), Kref) < TOL
    assert _rel(g.cpu().numpy(), gref) < TOL


def test_fast_c4_rows(gpu):
    """Headline size N=1024,T=64,d=7 on the benchmark's synthetic particles: K rows and grad rows of a
    subsample against the C oracle (full oracle run would take ~30 s of 8 cores)."""
    from sigsvgd_amd import ops

    X, _ = O.synthetic_inputs(1024, 64, 7)
    Xg = X.to(gpu)
    K, g = ops.gram_fwd_bwd(Xg, Xg, 1.0, y_is_x=True)
    torch.cuda.synchronize()
    Kn, gn = K.cpu().numpy(), g.cpu().numpy()
    assert np.isfinite(Kn).all() and np.isfinite(gn).all()
    assert np.array_equal(Kn, Kn.T)
    for rows in [(0, 4), (511, 515), (1020, 1024)]:
        Kref, gref = C.gram_fwd_bwd(X.numpy(), X.numpy(), 1.0, 0, rows=rows)
        assert _rel(Kn[rows[0]:rows[1]], Kref) < TOL
        assert np.abs(gn[rows[0]:rows[1]] - gref).max() / np.abs(gref).max() < TOL
