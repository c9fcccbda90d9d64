"""GPU parity at the edges of the static work partition of the register-resident kernel (gram_fast.hip): launches with
fewer items than workgroups, one row / one column, row tiles that are not full, ranges that cross row tiles, and the
strided row tiles of the sharded partial solve -- against the fp64 oracle, via the C ABI."""
import numpy as np
import pytest
import torch

from oracle import c_oracle as C

pytestmark = pytest.mark.gpu

TOL = 1e-5
# two fp32-sweep solves of one pair that differ in orientation (the symmetric launch solves (i, j), the ordered one also
# (j, i)) or launch geometry agree to a few ulps PER ENTRY; both are within TOL of the fp64 oracle
SELF = 4e-6


def _paths(A, T, d, seed, scale=0.05):
    rng = np.random.default_rng(seed)
    return np.cumsum(scale * rng.standard_normal((A, T, d)), axis=1).astype(np.float32)


def _rel(a, b):
    return float(np.abs(np.asarray(a, np.float64) - b).max() / max(np.abs(b).max(), 1e-300))


def _relK(a, b):
    """K parity as north_star states it: max over entries of |K - K_ref| / |K_ref| (K > 0 always)"""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    # (round 4: plain relative error per entry -- rounds 2-3 floored the denominator at 0.1; the 1e-6 only keeps an exact zero
    #  out of it.  Pairs whose K is small against their grid are solved by the exact fp64 pass now: DESIGN.md section 3)
    return float((np.abs(a - b) / np.maximum(np.abs(b), 1e-6)).max())


@pytest.mark.parametrize("A,B", [(1, 1), (1, 9), (9, 1), (3, 5), (8, 300), (300, 8), (67, 263)])
def test_ordered_launch_shapes(gpu, A, B):
    """X != Y: items = row tiles x all columns; 263 columns x 9 tiles = 2367 items over 256 workgroups crosses tiles."""
    from sigsvgd_amd import ops

    T, d, h = 16, 3, 0.9
    X, Y = _paths(A, T, d, 11), _paths(B, T, d, 12)
    go = np.random.default_rng(13).standard_normal((A, B))
    Kref, gref = C.gram_fwd_bwd(X, Y, h, 0, grad_out=go)
    Xg, Yg = torch.as_tensor(X, device=gpu), torch.as_tensor(Y, device=gpu)
    K, g = ops.gram_fwd_bwd(Xg, Yg, 1.0 / h, grad_out=torch.as_tensor(go, device=gpu, dtype=torch.float32))
    Kf = ops.gram_fwd(Xg, Yg, 1.0 / h)
    torch.cuda.synchronize()
    assert _relK(K.cpu().numpy(), Kref) < TOL
    assert _relK(Kf.cpu().numpy(), Kref) < TOL
    assert _rel(g.cpu().numpy(), gref) < TOL


@pytest.mark.parametrize("N", [1, 2, 7, 8, 9, 63, 257])
def test_symmetric_launch_shapes(gpu, N):
    """Y is X: items = columns from the tile's first row on; N = 257 leaves a last tile with one row."""
    from sigsvgd_amd import ops

    T, d, h = 12, 2, 1.1
    X = _paths(N, T, d, 21)
    Kref, gref = C.gram_fwd_bwd(X, X, h, 0)  # first-slot gradient; Y is X only says each unordered pair is solved once
    Xg = torch.as_tensor(X, device=gpu)
    K, g = ops.gram_fwd_bwd(Xg, Xg, 1.0 / h, y_is_x=True)
    Kf = ops.gram_fwd(Xg, Xg, 1.0 / h, y_is_x=True)
    torch.cuda.synchronize()
    assert _relK(K.cpu().numpy(), Kref) < TOL
    assert _relK(Kf.cpu().numpy(), Kref) < TOL
    assert _rel(g.cpu().numpy(), gref) < TOL
    assert torch.equal(K, K.T)


@pytest.mark.parametrize("fold", [False, True])
@pytest.mark.parametrize("N,stride,T", [(20, 3, 20), (70, 8, 20), (9, 2, 20), (5, 4, 20), (100, 3, 40), (131, 4, 64)])
def test_partial_shares_sum_to_full(gpu, N, stride, T, fold):
    """Strided row tiles (more ranks than tiles included), cyclic and folded ownership: the shares add up to the symmetric
    solve, every pair belongs to exactly one share, and a share holds exactly the tiles `ops.owned_tiles` lists."""
    from sigsvgd_amd import ops

    d, h = 7, 1.0
    X = _paths(N, T, d, 31)
    Xg = torch.as_tensor(X, device=gpu)
    K, g = ops.gram_fwd_bwd(Xg, Xg, 1.0 / h, y_is_x=True)
    Ks = torch.zeros_like(K)
    gs = torch.zeros(N, T, d, device=gpu, dtype=torch.float64)
    nw = ops.sym_tile_rows(T, d)
    ntile = (N + nw - 1) // nw
    for r in range(stride):
        Kp, gp = ops.gram_sym_partial(Xg, 1.0 / h, r, stride, fold=fold)
        Ks += Kp
        gs += gp
        # upper-triangle rows with an entry right of the diagonal are the rows of the owned tiles
        up = torch.triu(Kp != 0)
        rows = set(int(i) // nw for i in torch.nonzero(up.any(dim=1)).flatten().tolist())
        assert rows == set(ops.owned_tiles(ntile, r, stride, fold)), (r, rows)
    torch.cuda.synchronize()
    assert torch.equal(Ks, K)
    assert _relK(Ks.cpu().numpy(), K.double().cpu().numpy()) < SELF
    assert _rel(gs.cpu().numpy(), g.double().cpu().numpy()) < 1e-5


def test_paths_beyond_128_points(gpu):
    """T > 128 (dyadic order 0) is the coverage kernel's: its long-path layout (fp64 increments per band of 64 rows, S in the
    launch's scratch; round 4) takes paths while 64 (T-1) + 2 T d doubles fit 160 KB of LDS -- T = 190 with the gradient,
    which round 3 refused --, longer ones are refused loudly."""
    from sigsvgd_amd import ops

    A, B, h, d = 5, 4, 1.2, 2
    X, Y = _paths(A, 136, d, 41, scale=0.03), _paths(B, 136, d, 42, scale=0.03)
    Kref, gref = C.gram_fwd_bwd(X, Y, h, 0)
    K, g = ops.gram_fwd_bwd(torch.as_tensor(X, device=gpu), torch.as_tensor(Y, device=gpu), 1.0 / h)
    Ksr, gsr = C.gram_fwd_bwd(X, X, h, 0)
    Xg = torch.as_tensor(X, device=gpu)
    Ks, gs = ops.gram_fwd_bwd(Xg, Xg, 1.0 / h, y_is_x=True)
    torch.cuda.synchronize()
    assert _relK(K.cpu().numpy(), Kref) < TOL and _rel(g.cpu().numpy(), gref) < TOL
    assert _relK(Ks.cpu().numpy(), Ksr) < TOL and _rel(gs.cpu().numpy(), gsr) < TOL

    X2, Y2 = _paths(3, 190, d, 43, scale=0.03), _paths(4, 190, d, 44, scale=0.03)
    Kref2, gref2 = C.gram_fwd_bwd(X2, Y2, h, 0)
    X2g, Y2g = torch.as_tensor(X2, device=gpu), torch.as_tensor(Y2, device=gpu)
    K2 = ops.gram_fwd(X2g, Y2g, 1.0 / h)
    assert _relK(K2.cpu().numpy(), Kref2) < TOL
    K3, g3 = ops.gram_fwd_bwd(X2g, Y2g, 1.0 / h)
    assert _relK(K3.cpu().numpy(), Kref2) < TOL and _rel(g3.cpu().numpy(), gref2) < TOL
    X4 = torch.as_tensor(_paths(3, 300, d, 45, scale=0.03), device=gpu)
    with pytest.raises(RuntimeError, match="LDS"):
        ops.gram_fwd_bwd(X4, X4.clone(), 1.0 / h)
    with pytest.raises(RuntimeError, match="LDS"):
        ops.gram_fwd(X4, X4.clone(), 1.0 / h)
