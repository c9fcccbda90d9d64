"""CPU tests of the ORACLE itself: known-answer tests that stand in for the missing reference
fixtures of the PDE arithmetic, replay of the reference-generated fixtures for the
parts that live in the reference tree, and numpy-vs-C agreement."""
import numpy as np
import pytest
from scipy.special import i0

from helpers import golden
from oracle import c_oracle as C
from oracle import sigkernel_oracle as O


def _paths(A, T, d, seed, scale=0.3):
    rng = np.random.default_rng(seed)
    return np.cumsum(scale * rng.standard_normal((A, T, d)), axis=1)


# ---- (1) closed form: linear static kernel, one-segment paths -> I0(2 sqrt(<a,b>)) -----------------
def test_closed_form_bessel():
    a, b = np.array([0.7, 0.3]), np.array([0.5, 0.9])
    X = np.stack([np.zeros(2), a])[None]
    Y = np.stack([np.zeros(2), b])[None]
    exact = i0(2 * np.sqrt(a @ b))
    errs = [abs(O.gram(X, Y, O.LINEAR, 1.0, n)[0, 0] - exact) for n in (0, 2, 4, 6)]
    assert errs[0] < 8e-3 and errs[1] < 3e-4 and errs[2] < 3e-5 and errs[3] < 2e-6
    assert all(e2 < e1 for e1, e2 in zip(errs, errs[1:]))  # converges with the dyadic order
    assert abs(O.gram(X, Y, O.LINEAR, 1.0, 2)[0, 0] - 1.723261) < 1e-6  # value quoted in SURVEY.md §8c


# ---- (2) truncated signature inner product (independent Chen-identity implementation) --------------
def _signature(path, depth):
    """levels 1..depth of the signature of a piecewise-linear path via Chen's identity"""
    d = path.shape[1]
    sig = [np.zeros((d,) * k) for k in range(1, depth + 1)]
    for inc in np.diff(path, axis=0):
        seg = [inc]
        for k in range(2, depth + 1):
            seg.append(np.multiply.outer(seg[-1], inc) / k)
        new = []
        for k in range(1, depth + 1):
            acc = sig[k - 1] + seg[k - 1]
            for m in range(1, k):
                acc = acc + np.multiply.outer(sig[m - 1], seg[k - m - 1])
            new.append(acc)
        sig = new
    return sig


def test_truncated_signature_limit():
    x, y = _paths(1, 5, 2, 1, 0.25)[0], _paths(1, 5, 2, 2, 0.25)[0]
    depth = 8
    sx, sy = _signature(x, depth), _signature(y, depth)
    exact = 1.0 + sum(float((a * b).sum()) for a, b in zip(sx, sy))
    k6 = O.gram(x[None], y[None], O.LINEAR, 1.0, 6)[0, 0]
    assert abs(k6 - exact) < 5e-6


# ---- (3)(4) invariances -------------------------------------------------------------------------
@pytest.mark.parametrize("kind", [O.RBF, O.LINEAR])
def test_constant_path_symmetry_boundary(kind):
    X = _paths(4, 6, 3, 3)
    const = np.repeat(_paths(1, 1, 3, 4), 6, axis=1)
    assert np.allclose(O.gram(X, const, kind, 1.3, 2), 1.0, atol=1e-14)
    K = O.gram(X, X, kind, 1.3, 1)
    assert np.allclose(K, K.T, rtol=1e-13)
    Kfull = O.gram_forward_full(X, X, kind, 1.3, 1)[0]
    assert np.all(Kfull[..., 0, :] == 1.0) and np.all(Kfull[..., :, 0] == 1.0)


def test_repeated_point_invariance():
    X, Y = _paths(2, 6, 2, 5), _paths(3, 6, 2, 6)
    Xr = np.concatenate([X[:, :3], X[:, 2:3], X[:, 3:]], axis=1)  # repeat point 2: zero increments
    Yr = np.concatenate([Y[:, :3], Y[:, 2:3], Y[:, 3:]], axis=1)
    assert np.allclose(O.gram(X, Y, O.RBF, 0.8, 0), O.gram(Xr, Yr, O.RBF, 0.8, 0), rtol=1e-13)


# ---- (5) backward machinery -----------------------------------------------------------------------
def test_naive_stencil_gg_is_exact_adjoint():
    """with the first-order stencil GG == dK/dg, so the reference-style gradient equals finite
    differences of the forward to FD accuracy"""
    X, Y = _paths(3, 5, 2, 7), _paths(2, 5, 2, 8)
    _, g = O.gram_backward(X, Y, None, O.RBF, 2.0, 1, naive=True)
    num = np.zeros_like(X)
    eps = 1e-6
    for idx in np.ndindex(X.shape):
        Xp, Xm = X.copy(), X.copy()
        Xp[idx] += eps
        Xm[idx] -= eps
        num[idx] = (O.gram(Xp, Y, O.RBF, 2.0, 1, True).sum() - O.gram(Xm, Y, O.RBF, 2.0, 1, True).sum()) / (2 * eps)
    assert np.abs(num - g).max() / np.abs(g).max() < 1e-7


@pytest.mark.parametrize("n", [0, 1, 2])
def test_closed_form_backward_equals_recalled_fd_assembly(n):
    """the closed-form chain rule reproduces upstream's finite-difference Diff_1/Diff_2/grad_points
    assembly (to FD noise ~1e-7), also with a non-trivial grad_output"""
    X, Y = _paths(4, 7, 3, 9), _paths(5, 7, 3, 10)
    go = np.random.default_rng(0).standard_normal((4, 5))
    K1, g1 = O.gram_backward(X, Y, go, O.RBF, 2.0, n)
    K2, g2 = O.gram_backward_fd_literal(X, Y, go, 2.0, n)
    assert np.array_equal(K1, K2)
    assert np.abs(g1 - g2).max() / np.abs(g1).max() < 2e-6


def test_default_stencil_gradient_converges_first_order():
    """GG is NOT the adjoint of the second-order stencil; the mismatch shrinks ~1/r"""
    X, Y = _paths(2, 5, 2, 11), _paths(2, 5, 2, 12)
    errs = []
    for n in (0, 2, 4):
        _, g = O.gram_backward(X, Y, None, O.RBF, 2.0, n)
        num = np.zeros_like(X)
        eps = 1e-6
        for idx in np.ndindex(X.shape):
            Xp, Xm = X.copy(), X.copy()
            Xp[idx] += eps
            Xm[idx] -= eps
            num[idx] = (O.gram(Xp, Y, O.RBF, 2.0, n).sum() - O.gram(Xm, Y, O.RBF, 2.0, n).sum()) / (2 * eps)
        errs.append(np.abs(num - g).max() / np.abs(num).max())
    assert errs[0] > errs[1] > errs[2] and errs[2] < 0.1


def test_sym_flag_weights():
    X = _paths(4, 5, 2, 13)
    go = np.random.default_rng(1).standard_normal((4, 4))
    _, gs = O.gram_backward(X, X, go, O.RBF, 1.0, 1, sym=True)
    _, g2 = O.gram_backward(X, X, go + go.T, O.RBF, 1.0, 1)
    assert np.allclose(gs, g2)


def test_sweep_vectorised_equals_scalar_loop():
    g = O.refine(O.increments(O.static_gram(_paths(1, 6, 2, 14), _paths(1, 6, 2, 15), O.RBF, 1.0)), 2)[0, 0]
    for naive in (False, True):
        assert np.array_equal(O.pde_sweep(g, naive), O.pde_sweep_scalar(g, naive))


# ---- fixtures produced by the reference's own code ---------------------------------------------------
def test_static_kernel_matches_reference_fixture():
    G = golden()
    X, Y = G["sk_X"], G["sk_Y"]
    assert np.allclose(O.static_gram(X, Y, O.RBF, 0.7), G["sk_gram_h0.7"], rtol=1e-13, atol=0)
    assert np.allclose(O.static_batch(X, Y[:3], O.RBF, 0.7), G["sk_batch_h0.7"], rtol=1e-13, atol=0)
    assert np.allclose(O.static_gram(X, Y, O.RBF, 1.3), G["sk_gram_given_h"], rtol=1e-13, atol=0)
    h = O.bw_median(O.pairwise_sqdist(X, Y))
    assert np.allclose(O.static_gram(X, Y, O.RBF, h), G["sk_gram_median"], rtol=1e-12, atol=0)


def test_bw_median_matches_reference_fixture():
    G = golden()
    assert np.isclose(O.bw_median(G["bw_in"]), float(G["bw_out"]), rtol=1e-14)
    assert np.isclose(O.bw_median(G["bw_in"], 2.0), float(G["bw_out_scale2"]), rtol=1e-14)


def test_svgd_step_matches_reference_fixture():
    G = golden()
    Xn, v, _ = O.svgd_step_manual(G["svgd_X0"], G["svgd_step_in_score"], G["svgd_step_in_K"], G["svgd_step_in_gk"], 0.25)
    assert np.allclose(Xn, G["svgd_step_out_X"], rtol=1e-5, atol=1e-6)
    assert np.allclose(v, G["svgd_step_out_grad"], rtol=1e-5, atol=1e-6)
    vm = O.svgd_velocity(G["svgd_step_in_K"], G["svgd_step_in_score"], G["svgd_step_in_gk"], G["tsvgd_mask"])
    assert np.allclose(vm, G["tsvgd_velocity"], rtol=1e-5, atol=1e-6)


def test_reference_wiring_fixture_is_consistent_with_oracle():
    """K / grad_k captured through the reference's SignatureKernel + ScoreEstimator (fp64 upcast,
    fp32 cast-back) equal the oracle on the same particles"""
    G = golden()
    X = G["c1_X"]
    K, g = O.gram_backward(X, X, None, O.RBF, 1.5, 2)
    assert np.abs(G["c1_K"] - K).max() / np.abs(K).max() < 2e-7
    assert np.abs(G["c1_gradk"] - g).max() / np.abs(g).max() < 2e-7
    assert np.abs(G["c1_score_gradk"] - g).max() / np.abs(g).max() < 2e-7          # scheduler value 1
    assert np.abs(G["c1_score_gradk_2nd"] - g / np.sqrt(2)).max() / np.abs(g).max() < 2e-7


# ---- C restatement ------------------------------------------------------------------------------------
@pytest.mark.parametrize("kind,n,naive", [(0, 0, False), (0, 2, False), (0, 1, True), (1, 1, False)])
def test_c_oracle_matches_numpy_oracle(kind, n, naive):
    X = _paths(6, 9, 3, 20).astype(np.float32)
    Y = _paths(5, 9, 3, 21).astype(np.float32)
    go = np.random.default_rng(2).standard_normal((6, 5))
    K1, g1 = O.gram_backward(X, Y, go, kind, 1.7, n, naive)
    K2, g2 = C.gram_fwd_bwd(X, Y, 1.7, n, naive, kind, go)
    assert np.allclose(K1, K2, rtol=1e-12) and np.allclose(g1, g2, rtol=1e-10, atol=1e-13)
    K3, g3 = C.gram_fwd_bwd(X, Y, 1.7, n, naive, kind, go[2:4], rows=(2, 4))
    assert np.allclose(K3, K1[2:4], rtol=1e-12) and np.allclose(g3, g1[2:4], rtol=1e-10, atol=1e-13)


def test_c_oracle_update():
    rng = np.random.default_rng(3)
    K, s, gk, X = rng.standard_normal((7, 7)), rng.standard_normal((7, 4, 2)), rng.standard_normal((7, 4, 2)), rng.standard_normal((7, 4, 2))
    phi, Xn = C.svgd_update(K, s, gk, X, 0.1)
    assert np.allclose(phi, -O.svgd_velocity(K, s, gk)) and np.allclose(Xn, X + 0.1 * phi)
