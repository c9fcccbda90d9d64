"""GPU parity: long-path kernel (n=0, 65 <= T <= 128: 2 x 2 quadrants of the register-resident scheme, stored
forward solution -- csrc/gram_quad.hip) vs the fp64 oracle, via the C ABI."""
import numpy as np
import pytest
import torch

from oracle import c_oracle as C
from oracle import sigkernel_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-5
# two fp32-sweep solves of one pair that differ in orientation (the symmetric launch solves (i, j), the ordered one also
# (j, i)) or launch geometry agree to a few ulps PER ENTRY; both are within TOL of the fp64 oracle
SELF = 4e-6


def _paths(A, T, d, seed, scale=0.05):
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


@pytest.mark.parametrize("A,B,T,d", [(5, 6, 128, 14), (3, 9, 128, 7), (6, 5, 65, 3), (4, 4, 66, 2),
                                     (7, 3, 100, 7), (2, 5, 127, 16), (9, 2, 97, 1)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_long_fwd_bwd(gpu, A, B, T, d, dtype):
    from sigsvgd_amd import ops

    X, Y = _paths(A, T, d, 1), _paths(B, T, d, 2)
    h = 1.1
    go = np.random.default_rng(3).standard_normal((A, B)).astype(np.float32)
    Kref, gref = C.gram_fwd_bwd(X, Y, h, 0, grad_out=go.astype(np.float64))
    Xg, Yg, gog = (torch.as_tensor(t, device=gpu).to(dtype) for t in (X, Y, go))
    K1 = ops.gram_fwd(Xg, Yg, 1.0 / h)
    K2, g2 = ops.gram_fwd_bwd(Xg, Yg, 1.0 / h, grad_out=gog)
    torch.cuda.synchronize()
    assert _relK(K1.cpu().numpy(), Kref) < TOL
    assert _relK(K2.cpu().numpy(), Kref) < TOL
    assert _rel(g2.cpu().numpy(), gref) < TOL
    if T * d <= 128 * 14:  # the coverage kernel's compact layout tops out at T=128, d=14 (LDS)
        K3, g3 = ops.gram_fwd_bwd(Xg, Yg, 1.0 / h, grad_out=gog, force_generic=True)
        assert _rel(g2.cpu().numpy(), g3.double().cpu().numpy()) < TOL


def test_long_self_gram_c5_shape(gpu):
    """C5 path shape (T=128, d=14) on the benchmark's synthetic particles, Y is X (sym weighting too)"""
    from sigsvgd_amd import ops

    X, _ = O.synthetic_inputs(24, 128, 14)
    Xg = X.to(gpu)
    K, g = ops.gram_fwd_bwd(Xg, Xg, 1.0, y_is_x=True)
    Kref, gref = C.gram_fwd_bwd(X.numpy(), X.numpy(), 1.0, 0)
    assert _relK(K.cpu().numpy(), Kref) < TOL and _rel(g.cpu().numpy(), gref) < TOL
    K2, g2 = ops.gram_fwd_bwd(Xg, Xg, 1.0, sym=True, y_is_x=True)
    assert _rel(g2.cpu().numpy(), 2 * gref) < TOL


@pytest.mark.parametrize("N,T,d", [(9, 128, 14), (13, 65, 3), (6, 100, 7), (5, 127, 16), (1, 96, 2), (17, 128, 1)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_long_symmetric_solve_equals_ordered_pairs(gpu, N, T, d, dtype):
    """y_is_x=True solves every unordered pair once (mirrored K, column-side gradient through the
    travelling accumulators); with asymmetric weights it must still equal the ordered-pair result."""
    from sigsvgd_amd import ops

    X = _paths(N, T, d, 5)
    h = 0.9
    go = np.random.default_rng(6).standard_normal((N, N)).astype(np.float32)
    Kref, gref = C.gram_fwd_bwd(X, X, h, 0, grad_out=go.astype(np.float64))
    Xg, gog = torch.as_tensor(X, device=gpu).to(dtype), torch.as_tensor(go, device=gpu).to(dtype)
    K, g = ops.gram_fwd_bwd(Xg, Xg, 1.0 / h, grad_out=gog, y_is_x=True)
    assert _relK(K.cpu().numpy(), Kref) < TOL and _rel(g.cpu().numpy(), gref) < TOL
    assert torch.equal(K, K.T)  # mirrored stores
    Ko, g_o = ops.gram_fwd_bwd(Xg, Xg, 1.0 / h, grad_out=gog)  # ordered pairs on the same kernel family
    assert _relK(K.cpu().numpy(), Ko.double().cpu().numpy()) < SELF
    assert _rel(g.cpu().numpy(), g_o.double().cpu().numpy()) < 1e-5
    # ones weights, and the sym=True weighting (grad_out symmetrised)
    K1, g1 = ops.gram_fwd_bwd(Xg, Xg, 1.0 / h, y_is_x=True)
    _, gref1 = C.gram_fwd_bwd(X, X, h, 0)
    assert _rel(g1.cpu().numpy(), gref1) < TOL
    _, g2 = ops.gram_fwd_bwd(Xg, Xg, 1.0 / h, grad_out=gog, sym=True, y_is_x=True)
    _, gref2 = C.gram_fwd_bwd(X, X, h, 0, grad_out=(go + go.T).astype(np.float64))
    assert _rel(g2.cpu().numpy(), gref2) < TOL


@pytest.mark.parametrize("fold", [False, True])
@pytest.mark.parametrize("N,T,d,world", [(22, 128, 14, 3), (10, 70, 5, 2), (9, 96, 3, 4), (40, 80, 16, 2)])
def test_long_partials_sum_to_full(gpu, N, T, d, world, fold):
    """sigsvgd_gram_sym_partial on the long-path shapes: the per-rank partials (4-row tiles, cyclic) add up
    to the full symmetric solve, and each owned pair appears in exactly one partial."""
    from sigsvgd_amd import ops

    X = _paths(N, T, d, 8)
    Xg = torch.as_tensor(X, device=gpu)
    go = torch.as_tensor(np.random.default_rng(9).uniform(0.5, 1.5, (N, N)).astype(np.float32), device=gpu)
    Kf, gf = ops.gram_fwd_bwd(Xg, Xg, 1.0, grad_out=go, y_is_x=True)
    Ks = torch.zeros_like(Kf)
    gs = torch.zeros(N, T, d, dtype=torch.float64, device=gpu)
    cover = torch.zeros(N, N, device=gpu)
    for r in range(world):
        Kp, gp = ops.gram_sym_partial(Xg, 1.0, r, world, grad_out=go, fold=fold)
        Ks += Kp
        gs += gp
        cover += (Kp != 0).float()
    assert torch.equal(cover, torch.ones_like(cover))
    assert torch.equal(Ks, Kf)
    assert _rel(gs.cpu().numpy(), gf.double().cpu().numpy()) < 1e-6
    Kref, gref = C.gram_fwd_bwd(X, X, 1.0, 0, grad_out=go.double().cpu().numpy())
    assert _rel(gs.cpu().numpy(), gref) < TOL


def test_long_symmetric_large_property(gpu):
    """C5 path shape at N=96: K symmetric with unit-free diagonal structure, and the symmetric solve equals
    the ordered solve (no oracle at this size)."""
    from sigsvgd_amd import ops

    X, _ = O.synthetic_inputs(96, 128, 14)
    Xg = X.to(gpu)
    K, g = ops.gram_fwd_bwd(Xg, Xg, 1.0, y_is_x=True)
    Ko, g_o = ops.gram_fwd_bwd(Xg, Xg.clone(), 1.0)
    assert torch.equal(K, K.T)
    assert _relK(K.cpu().numpy(), Ko.double().cpu().numpy()) < SELF
    assert _rel(g.cpu().numpy(), g_o.double().cpu().numpy()) < 1e-5
    assert torch.isfinite(g).all()


@pytest.mark.parametrize("A,B,T,d,sym,scale", [(2, 3, 128, 14, False, 0.05), (5, 5, 128, 14, True, 0.05), (3, 2, 65, 3, False, 0.05),
                                               (4, 4, 100, 7, True, 0.05), (6, 6, 66, 2, True, 0.05), (4, 4, 128, 14, True, 0.15),
                                               (3, 5, 97, 5, False, 0.3), (9, 9, 127, 8, True, 0.05), (3, 3, 65, 16, True, 0.05),
                                               (10, 10, 129 - 1, 9, True, 0.1)])
def test_stored_forward_quadrant_kernel(gpu, A, B, T, d, sym, scale):
    """gram_quad.hip: the long-path kernel (stored forward solution), on smooth AND
    rough paths (scale 0.15 / 0.3: increments far beyond what a regenerating kernel could accept), ordered and symmetric,
    against the C oracle; K also from the forward-only launch of the default kernel."""
    from sigsvgd_amd import ops

    rng = np.random.default_rng(7)
    X = np.cumsum(scale * rng.standard_normal((A, T, d)), axis=1).astype(np.float32)
    Y = X if sym else np.cumsum(scale * rng.standard_normal((B, T, d)), axis=1).astype(np.float32)
    Kref, gref = C.gram_fwd_bwd(X, Y, 1.0, 0)
    Xg, Yg = torch.as_tensor(X, device=gpu), torch.as_tensor(Y, device=gpu)
    K, g = ops.gram_fwd_bwd(Xg, Xg if sym else Yg, 1.0, y_is_x=sym, stored_forward=True)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(g).all())
    assert _relK(K.cpu().numpy(), Kref) < TOL and _rel(g.cpu().numpy(), gref) < TOL
    # forward-only launch of the same kernel, and fp64 I/O
    Kf = ops.gram_fwd(Xg, Xg if sym else Yg, 1.0, y_is_x=sym, stored_forward=True)
    assert _relK(Kf.cpu().numpy(), Kref) < TOL
    K64, g64 = ops.gram_fwd_bwd(Xg.double(), (Xg if sym else Yg).double(), 1.0, y_is_x=sym, stored_forward=True)
    assert K64.dtype == torch.float64 and _relK(K64.cpu().numpy(), Kref) < TOL and _rel(g64.cpu().numpy(), gref) < TOL
    if sym:
        assert np.array_equal(K.cpu().numpy(), K.cpu().numpy().T)
        # weighted backward (grad_out) and the sharded partial solve: two shares sum to the full result
        go = torch.as_tensor(rng.standard_normal((A, A)).astype(np.float32), device=gpu)
        Kw, gw = ops.gram_fwd_bwd(Xg, Xg, 1.0, grad_out=go, y_is_x=True, stored_forward=True)
        _, gwref = C.gram_fwd_bwd(X, X, 1.0, 0, grad_out=go.double().cpu().numpy())
        assert _rel(gw.cpu().numpy(), gwref) < TOL
        if True:
            Ks = torch.zeros_like(K)
            gs = torch.zeros((A, T, d), dtype=torch.float64, device=gpu)
            for off in range(2):
                Kp, gp = ops.gram_sym_partial(Xg, 1.0, off, 2)
                Ks += Kp
                gs += gp
            assert torch.equal(Ks, K) and _rel(gs.cpu().numpy(), gref) < TOL


@pytest.mark.parametrize("T,d,scale,h", [(100, 2, 0.2, 0.1), (100, 2, 0.5, 1.0), (128, 2, 0.2, 1.0), (80, 2, 0.2, 0.1)])
def test_long_oscillating_solutions_per_entry(gpu, T, d, scale, h):
    """Rough paths in few channels: the discrete solution oscillates and K[P][P] can be a small remainder of much larger
    values on the pair's grid (fp32 sweeps alone: up to 2.7e-5 per entry at T = 100, d = 2).  The quadrant kernel flags the
    pairs with max |K_grid| > 4 max(|K|, 0.1) and the coverage kernel solves their K again in fp64 (gram_generic.hip,
    generic_repair_launch): every entry inside the tolerance in every launch form, the sharded partial solve included."""
    from sigsvgd_amd import ops

    N = 10
    rng = np.random.default_rng(0)
    X = np.cumsum(scale * rng.standard_normal((N, T, d)), axis=1).astype(np.float32)
    Kref, gref = C.gram_fwd_bwd(X, X, h, 0)
    assert Kref.min() < 0.5  # (the regime the test is about: solutions that cancel)
    Xg = torch.as_tensor(X, device=gpu)
    for K, g in [ops.gram_fwd_bwd(Xg, Xg, 1.0 / h, y_is_x=True), ops.gram_fwd_bwd(Xg, Xg.clone(), 1.0 / h)]:
        assert _relK(K.cpu().numpy(), Kref) < 8e-6
        assert _rel(g.cpu().numpy(), gref) < TOL
    for K in [ops.gram_fwd(Xg, Xg, 1.0 / h, y_is_x=True), ops.gram_fwd(Xg, Xg.clone(), 1.0 / h)]:
        assert _relK(K.cpu().numpy(), Kref) < 8e-6
    Ks = torch.zeros(N, N, device=gpu)
    for r in range(2):
        Kp, _ = ops.gram_sym_partial(Xg, 1.0 / h, r, 2, fold=True)
        Ks += Kp
    assert _relK(Ks.cpu().numpy(), Kref) < 8e-6
