"""GPU parity of the vector kernels (Gaussian / scaled Gaussian / IMQ / scaled IMQ), the truncated
signature and PathSigKernel against the fp64 oracle and the reference-generated fixtures
(tests/golden/ref_vector_kernels.npz).  Tolerance: fp32 I/O -> 1e-5 relative to max-abs; fp64 I/O -> 1e-11."""
import os

import numpy as np
import pytest
import torch

from oracle import vector_oracle as VO

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_vector_kernels.npz"))


def rel(a, b):
    a = a.detach().double().cpu().numpy() if hasattr(a, "detach") else np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def t64(a, gpu):
    return torch.as_tensor(np.asarray(a), dtype=torch.float64, device=gpu)


# ---- reference fixtures replayed through the drop-in classes (fp64 in, like the fixtures) ---------------
@pytest.mark.parametrize("name", ["gauss", "imq"])
def test_plain_kernels_reference_fixtures(gpu, name):
    from sigsvgd_amd.kernels import GaussianKernel, IMQKernel

    ker = {"gauss": GaussianKernel, "imq": IMQKernel}[name]()
    X, Y = t64(G["X"], gpu), t64(G["Y"], gpu)
    K, dK = ker(X, Y, h=0.8)
    assert K.dtype == torch.float64 and K.shape == (7, 7) and dK.shape == (7, 5)
    assert rel(K, G[f"{name}_h0.8_K"]) < 1e-11 and rel(dK, G[f"{name}_h0.8_dK"]) < 1e-11
    assert rel(ker(X, Y, h=0.8, compute_grad=False), G[f"{name}_h0.8_Konly"]) < 1e-11
    K, dK = ker(X, Y)  # median heuristic (float32 log inside, as in the reference)
    assert rel(K, G[f"{name}_med_K"]) < 1e-6 and rel(dK, G[f"{name}_med_dK"]) < 1e-6
    K, dK = ker(X, X, h=1.1)
    assert rel(K, G[f"{name}_xx_K"]) < 1e-11 and rel(dK, G[f"{name}_xx_dK"]) < 1e-11


@pytest.mark.parametrize("name", ["sgauss", "simq"])
def test_scaled_kernels_reference_fixtures(gpu, name):
    from sigsvgd_amd.kernels import ScaledGaussianKernel, ScaledIMQKernel

    ker = {"sgauss": ScaledGaussianKernel, "simq": ScaledIMQKernel}[name]()
    X, Y, M, Mns = (t64(G[k], gpu) for k in ("X", "Y", "M", "Mns"))
    for key, kw in [("I_h0.8", dict(h=0.8)), ("M_h0.8", dict(M=M, h=0.8)), ("Mns_h1.3", dict(M=Mns, h=1.3))]:
        K, dK = ker(X, Y, **kw)
        assert rel(K, G[f"{name}_{key}_K"]) < 1e-11, key
        assert rel(dK, G[f"{name}_{key}_dK"]) < 1e-11, key
    K, dK = ker(X, Y, M=M)
    assert rel(K, G[f"{name}_M_med_K"]) < 1e-6 and rel(dK, G[f"{name}_M_med_dK"]) < 1e-6


def test_three_dimensional_particles_fixture(gpu):
    from sigsvgd_amd.kernels import GaussianKernel

    X3 = t64(G["X3"], gpu)
    K, dK = GaussianKernel()(X3, X3, h=1.7)
    assert dK.shape == (6, 12)
    assert rel(K, G["gauss_X3_K"]) < 1e-11 and rel(dK, G["gauss_X3_dK"]) < 1e-11


# ---- fp32 against the fp64 oracle at awkward and larger sizes -----------------------------------------
@pytest.mark.parametrize("A,B,D", [(1, 1, 1), (3, 70, 5), (65, 17, 67), (130, 130, 129), (257, 300, 448)])
@pytest.mark.parametrize("kind", ["gaussian", "imq"])
def test_vec_ops_fp32_vs_oracle(gpu, A, B, D, kind):
    from sigsvgd_amd import _lib, ops

    rng = np.random.default_rng(A * 1000 + B + D)
    X = rng.normal(size=(A, D)).astype(np.float32)
    Y = (rng.normal(size=(B, D)) * 0.9 + 0.1).astype(np.float32)
    go = rng.uniform(0.5, 1.5, size=(A, B)).astype(np.float32)
    h = float(np.sqrt(D))
    sq = ops.vec_sqdist(torch.as_tensor(X, device=gpu), torch.as_tensor(Y, device=gpu))
    want_sq = VO.pw_dist_sq(X, Y)
    assert rel(sq, want_sq) < 1e-5
    k = _lib.VEC_GAUSSIAN if kind == "gaussian" else _lib.VEC_IMQ
    K, dK = ops.vec_kernel(sq, torch.as_tensor(X, device=gpu), torch.as_tensor(Y, device=gpu), k, 1 / h**2, -1 / h**2,
                           grad_out=torch.as_tensor(go, device=gpu))
    wantK = np.exp(-0.5 / h**2 * want_sq) if kind == "gaussian" else (1 + 0.5 * want_sq / h**2) ** -0.5
    assert rel(K, wantK) < 1e-5
    want = VO.vec_kernel_weighted_grad(want_sq, X, Y, go, kind, h, -1 / h**2)
    assert rel(dK, want) < 1e-5
    # gradient only / kernel only
    assert ops.vec_kernel(sq, None, None, k, 1 / h**2, 0.0, want_grad=False)[1] is None
    K2, dK2 = ops.vec_kernel(sq, torch.as_tensor(X, device=gpu), torch.as_tensor(Y, device=gpu), k, 1 / h**2, -1 / h**2,
                             grad_out=torch.as_tensor(go, device=gpu), want_K=False)
    assert K2 is None and torch.equal(dK2, dK)


def test_metric_sqdist_fp32_vs_oracle(gpu):
    from sigsvgd_amd import ops

    rng = np.random.default_rng(3)
    X, Y = rng.normal(size=(90, 33)).astype(np.float32), rng.normal(size=(75, 33)).astype(np.float32)
    R = rng.normal(size=(33, 33))
    M = (R @ R.T / 33 + np.eye(33)).astype(np.float32)
    Xg, Yg, Mg = (torch.as_tensor(a, device=gpu) for a in (X, Y, M))
    sq = ops.vec_sqdist(Xg, Yg, Xg @ Mg, Yg @ Mg)
    want, _ = VO.scaled_pw_dist_sq(X, Y, M)
    assert rel(sq, want) < 1e-5
    assert float(sq.min()) >= 0.0


def test_autograd_through_K_matches_torch(gpu):
    """compute_grad=False returns a differentiable K (reference ScoreEstimator._svgd_ag_score pattern),
    including the dependence of a median bandwidth on the inputs."""
    from sigsvgd_amd.kernels import GaussianKernel, ScaledIMQKernel

    g = torch.Generator().manual_seed(4)
    X0 = torch.randn(20, 6, generator=g, dtype=torch.float64)
    M = torch.randn(6, 6, generator=g, dtype=torch.float64)
    M = (M @ M.T / 6 + torch.eye(6, dtype=torch.float64))
    for ker, kw, f in [
        (GaussianKernel(), {}, lambda sq, h: (-0.5 / h**2 * sq).exp()),
        (GaussianKernel(bandwidth_fn=lambda _: 1.3), {}, lambda sq, h: (-0.5 / h**2 * sq).exp()),
        (ScaledIMQKernel(bandwidth_fn=lambda _: 0.9), {"M": M}, lambda sq, h: (1 + 0.5 * sq / h**2) ** -0.5),
    ]:
        x = X0.clone().to(gpu).requires_grad_(True)
        K = ker(x, x.detach(), compute_grad=False, **{k: v.to(gpu) for k, v in kw.items()})
        (gx,) = torch.autograd.grad(K.sum(), x)
        xc = X0.clone().requires_grad_(True)
        diff = xc[:, None, :] - xc.detach()[None, :, :]
        sq = ((diff @ kw["M"]) * diff).sum(-1) if kw else (diff * diff).sum(-1)
        Kc = f(sq, ker.get_bandwidth(sq))
        (gc,) = torch.autograd.grad(Kc.sum(), xc)
        assert rel(K, Kc.detach().numpy()) < 1e-6
        assert rel(gx, gc.numpy()) < 1e-6


def test_sqdist_autograd_second_slot(gpu):
    from sigsvgd_amd.kernels._kernels import _SqDist

    g = torch.Generator().manual_seed(9)
    X = torch.randn(9, 4, generator=g, dtype=torch.float64)
    Y = torch.randn(11, 4, generator=g, dtype=torch.float64)
    W = torch.randn(9, 11, generator=g, dtype=torch.float64)
    xg, yg = X.to(gpu).requires_grad_(True), Y.to(gpu).requires_grad_(True)
    sq = _SqDist.apply(xg, yg, None)
    gx, gy = torch.autograd.grad((sq * W.to(gpu)).sum(), (xg, yg))
    xc, yc = X.clone().requires_grad_(True), Y.clone().requires_grad_(True)
    sqc = ((xc[:, None] - yc[None]) ** 2).sum(-1)
    gxc, gyc = torch.autograd.grad((sqc * W).sum(), (xc, yc))
    assert rel(gx, gxc.numpy()) < 1e-12 and rel(gy, gyc.numpy()) < 1e-12


# ---- truncated signature ---------------------------------------------------------------------------------
@pytest.mark.parametrize("N,L,C,depth,bp", [(1, 2, 1, 1, False), (5, 8, 2, 3, True), (3, 25, 2, 3, True), (4, 6, 3, 4, False),
                                            (2, 64, 7, 3, True), (2, 5, 2, 6, True), (70, 3, 4, 2, True)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_signature_vs_oracle(gpu, N, L, C, depth, bp, dtype):
    from sigsvgd_amd import ops

    rng = np.random.default_rng(N + L * 7 + C * 31 + depth)
    X = np.cumsum(0.4 * rng.normal(size=(N, L, C)), axis=1)
    S = ops.signature(torch.as_tensor(X, dtype=dtype, device=gpu), depth, basepoint=bp)
    assert S.dtype == dtype and tuple(S.shape) == (N, VO.signature_channels(C, depth))
    assert ops.signature_channels(C, depth) == S.shape[1]
    want = VO.signature(X.astype(np.float32) if dtype == torch.float32 else X, depth, bp)
    assert rel(S, want) < (1e-6 if dtype == torch.float32 else 1e-13)


def test_signature_straight_line_and_single_point(gpu):
    from sigsvgd_amd import ops

    a = np.array([0.7, -0.3, 0.2])
    x = np.linspace(0, 1, 6)[:, None] * a
    S = ops.signature(torch.as_tensor(x[None], device=gpu), 3, basepoint=True)[0].cpu().numpy()
    want = np.concatenate([a, np.multiply.outer(a, a).ravel() / 2, np.multiply.outer(np.multiply.outer(a, a), a).ravel() / 6])
    np.testing.assert_allclose(S, want, rtol=1e-13, atol=1e-15)
    one = ops.signature(torch.ones(2, 1, 3, dtype=torch.float64, device=gpu), 2, basepoint=False)
    assert float(one.abs().max()) == 0.0  # a single point has no increments


def test_signature_too_large_for_lds_is_an_error(gpu):
    from sigsvgd_amd import ops

    with pytest.raises(RuntimeError, match="LDS"):
        ops.signature(torch.zeros(1, 4, 16, device=gpu), 4)  # 69,904 channels


def test_path_sig_kernel_fixture_and_oracle(gpu):
    from sigsvgd_amd.kernels import PathSigKernel

    P1, P2 = t64(G["P1"], gpu), t64(G["P2"], gpu)
    psk = PathSigKernel()
    K, dK = psk(P1, P2, depth=3, h=0.9)  # h is ignored, as in the reference
    assert K.shape == (6, 6) and dK.shape == (6, 14)
    assert rel(K, G["psk_d3_h0.9_K"]) < 1e-6 and rel(dK, G["psk_d3_h0.9_dK"]) < 1e-6
    K, dK = psk(P1, P2, depth=2)
    assert rel(K, G["psk_d2_med_K"]) < 1e-6 and rel(dK, G["psk_d2_med_dK"]) < 1e-6
    assert rel(psk(P1, P1, depth=3, compute_grad=False), G["psk_d3_Konly"]) < 1e-6
    # the reference's test shape (tests/test_traj_kernels.py: batch 128, 25 points, (cos, sin) channels), fp32
    g = torch.Generator().manual_seed(0)
    X = torch.randn(128, 25, 1, generator=g)
    Y = torch.randn(128, 25, 1, generator=g)
    phi = lambda t: torch.cat((t.cos(), t.sin()), -1)
    K, dK = psk(phi(X).to(gpu), phi(Y).to(gpu), X.to(gpu), depth=3, h=2.0**0.5)
    wK, wdK, _ = VO.path_sig_kernel(phi(X).numpy(), phi(Y).numpy(), depth=3)
    assert rel(K, wK) < 1e-5 and rel(dK, wdK) < 1e-5


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("N,L,C,depth,bp", [(5, 9, 2, 3, True), (4, 7, 3, 2, False), (3, 12, 2, 4, True), (6, 5, 4, 3, False),
                                            (3, 20, 7, 3, True), (2, 1, 3, 2, True), (3, 6, 1, 5, False),
                                            # the one-thread-per-path kernel of the small signatures (C = 2 at depth 2-4, C = 3 at depth 2-3,
                                            # C = 4, 5, 6 at depth 2), more paths than one workgroup of 64 threads
                                            (70, 6, 2, 2, False), (130, 5, 4, 2, True), (65, 4, 2, 3, False), (67, 3, 3, 2, True),
                                            (9, 7, 2, 4, True), (66, 5, 3, 3, False), (5, 6, 5, 2, True), (4, 9, 6, 2, False)])
def test_signature_backward_vs_oracle(gpu, N, L, C, depth, bp, dtype):
    """the HIP adjoint of the signature (`sigsvgd_signature_backward`, reached by autograd through `ops.signature`) against
    the oracle's reverse-mode gradient, which tests/test_oracle_vector.py pins with finite differences of the signature"""
    from sigsvgd_amd import ops

    rng = np.random.default_rng(100 * C + 10 * depth + L)
    X = np.cumsum(0.3 * rng.standard_normal((N, L, C)), axis=1)
    W = rng.standard_normal((N, VO.signature_channels(C, depth)))
    if dtype == torch.float32:
        X, W = X.astype(np.float32).astype(np.float64), W.astype(np.float32).astype(np.float64)
    sig_ref, g_ref = VO.signature_vjp(X, W, depth, bp)
    x = torch.as_tensor(X, dtype=dtype, device=gpu).requires_grad_(True)
    S = ops.signature(x, depth, basepoint=bp)
    (g,) = torch.autograd.grad((S * torch.as_tensor(W, dtype=dtype, device=gpu)).sum(), x)
    tol = 1e-11 if dtype == torch.float64 else 1e-5
    assert g.dtype == dtype and g.shape == x.shape
    assert rel(S.detach(), sig_ref) < tol and rel(g, g_ref) < tol
    # the explicit entry point gives the same bits, twice
    g2 = ops.signature_backward(x.detach(), torch.as_tensor(W, dtype=dtype, device=gpu), depth, bp)
    assert torch.equal(g, g2) and torch.equal(g2, ops.signature_backward(x.detach(), torch.as_tensor(W, dtype=dtype, device=gpu), depth, bp))
    # no graph, no node
    assert not ops.signature(x.detach(), depth, basepoint=bp).requires_grad


def test_path_sig_kernel_is_differentiable_through_the_signature(gpu):
    """PathSigKernel has analytic_grad=False (reference _traj_kernels.py:92): ScoreEstimator routes it to
    `k_xx = kernel(x, x.detach(), compute_grad=False); grad_k = autograd.grad(k_xx.sum(), x)` (score.py:50-55) and
    SVGD._compute_kernel does the same (svgd.py:41-43) -- both differentiate THROUGH signatory.signature.  Here: through
    the HIP signature and its HIP adjoint; checked against the oracle's chain (Gaussian on signatures, fixed bandwidth,
    then the signature's vjp) and against central finite differences of the oracle's K.sum() at depth 2 and 3."""
    from sigsvgd_amd.inference import SVGD, ScoreEstimator
    from sigsvgd_amd.kernels import GaussianKernel, PathSigKernel

    rng = np.random.default_rng(21)
    h = 1.7
    for depth in (2, 3):
        X = np.cumsum(0.4 * rng.standard_normal((7, 8, 2)), axis=1)
        psk = PathSigKernel(static_kernel=GaussianKernel(bandwidth_fn=lambda _: h))
        x = torch.as_tensor(X, dtype=torch.float64, device=gpu).requires_grad_(True)
        K = psk(x, x.detach(), depth=depth, compute_grad=False)
        (gk,) = torch.autograd.grad(K.sum(), x)

        def Ksum(Xa):  # oracle: first slot varies, second fixed (x.detach())
            Kc, _, _ = VO.gaussian(VO.signature(Xa, depth, True), VO.signature(X, depth, True), h)
            return Kc

        Kref = Ksum(X)
        assert rel(K.detach(), Kref) < 1e-10
        # oracle chain rule: dK_ij/dS_i = -(S_i - S_j)/h^2 K_ij, then the signature's vjp
        Sx = VO.signature(X, depth, True)
        dS = -((Sx[:, None, :] - Sx[None, :, :]) / h**2 * Kref[..., None]).sum(1)
        _, g_chain = VO.signature_vjp(X, dS, depth, True)
        assert rel(gk, g_chain) < 1e-9
        eps, fd = 1e-6, np.zeros_like(X)
        for idx in [(0, 0, 0), (3, 4, 1), (6, 7, 0), (2, 2, 1), (5, 0, 1)]:
            Xp, Xm = X.copy(), X.copy()
            Xp[idx] += eps
            Xm[idx] -= eps
            fd[idx] = (Ksum(Xp).sum() - Ksum(Xm).sum()) / (2 * eps)
            assert abs(float(gk[idx]) - fd[idx]) < 1e-6 * np.abs(g_chain).max()
    # the estimator and SVGD routes of the reference run end to end
    xs = torch.as_tensor(X, dtype=torch.float32, device=gpu).requires_grad_(True)
    psk32 = PathSigKernel(static_kernel=GaussianKernel(bandwidth_fn=lambda _: h))
    est = ScoreEstimator(psk32, lambda x: ((x**2).sum((1, 2)), {}), {})
    assert est.score.__func__ is ScoreEstimator._svgd_ag_score
    glp, aux = est.score(xs)
    assert aux["grad_k"].shape == xs.shape and bool(torch.isfinite(aux["grad_k"]).all())
    assert rel(aux["grad_k"], g_chain) < 1e-4 and rel(aux["k_xx"], Kref) < 1e-5
    k2, g2 = SVGD(psk32, optimizer_class=None)._compute_kernel(xs.detach().requires_grad_(True))
    assert rel(g2.reshape(xs.shape), g_chain) < 1e-4


def test_svgd_with_default_gaussian_kernel_matches_closed_form(gpu):
    """SVGD(kernel=None) uses GaussianKernel (reference svgd.py:24-25): one manual step == oracle."""
    from sigsvgd_amd.inference import SVGD

    rng = np.random.default_rng(12)
    X = rng.normal(size=(40, 6)).astype(np.float32)
    score = rng.normal(size=(40, 6)).astype(np.float32)
    s = SVGD(optimizer_class=None, lr=0.1)
    Xn, it = s.step(torch.as_tensor(X, device=gpu), torch.as_tensor(score, device=gpu), None)
    K, dK, _ = VO.gaussian(X, X)
    v = -((K @ score.astype(np.float64) - dK) / 40)
    assert rel(it["k_xx"], K) < 1e-5
    assert rel(Xn, X - 0.1 * v) < 1e-5


def test_trajectory_kernel_autograd_to_actions(gpu):
    from sigsvgd_amd.kernels import TrajectoryKernel

    g = torch.Generator().manual_seed(2)
    a0 = torch.randn(12, 5, generator=g, dtype=torch.float64)
    act = a0.clone().to(gpu).requires_grad_(True)
    tau = torch.cumsum(act, 1)  # a toy differentiable rollout
    K, dK = TrajectoryKernel()(tau, tau.detach(), act, h=1.4)
    ac = a0.clone().requires_grad_(True)
    tc = torch.cumsum(ac, 1)
    sq = ((tc[:, None] - tc.detach()[None]) ** 2).sum(-1)
    Kc = (-0.5 / 1.4**2 * sq).exp()
    (dc,) = torch.autograd.grad(Kc.sum(), ac)
    assert rel(K, Kc.detach().numpy()) < 1e-12 and rel(dK, dc.numpy()) < 1e-12


def test_vector_kernel_argument_errors(gpu):
    from sigsvgd_amd import _lib, ops
    from sigsvgd_amd.kernels import GaussianKernel

    x = torch.zeros(4, 3, device=gpu)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.vec_sqdist(torch.zeros(4, 3), torch.zeros(4, 3))
    with pytest.raises(ValueError):
        ops.vec_sqdist(x, torch.zeros(4, 2, device=gpu))
    with pytest.raises(RuntimeError, match="1/h"):
        ops.vec_kernel(torch.zeros(4, 4, device=gpu), x, x, _lib.VEC_GAUSSIAN, 0.0, 1.0)
    with pytest.raises(AssertionError):
        GaussianKernel()(x, torch.zeros(5, 3, device=gpu))
    with pytest.raises(ValueError):
        GaussianKernel(bandwidth_fn=3.0)


# ---- fused fixed-bandwidth kernel (fp32 MFMA) ----------------------------------------------------------------
@pytest.mark.parametrize("A,B,D", [(1, 1, 1), (3, 70, 5), (65, 17, 67), (130, 130, 129), (257, 300, 448), (700, 513, 64)])
@pytest.mark.parametrize("kind", ["gaussian", "imq"])
@pytest.mark.parametrize("offset", [0.0, 100.0])
def test_vec_kernel_fused_fp32_vs_oracle(gpu, A, B, D, kind, offset):
    """One launch for a given bandwidth (distance on the matrix cores, operands centred on Y[0]) against the fp64
    oracle; `offset` moves the whole cloud away from the origin, where an uncentred |x|^2 + |y|^2 - 2 x.y loses bits."""
    from sigsvgd_amd import _lib, ops

    rng = np.random.default_rng(A * 1000 + B + D)
    X = (rng.normal(size=(A, D)) + offset).astype(np.float32)
    Y = (rng.normal(size=(B, D)) * 0.9 + 0.1 + offset).astype(np.float32)
    go = rng.uniform(0.5, 1.5, size=(A, B)).astype(np.float32)
    h = float(np.sqrt(D))
    Xg, Yg, gog = (torch.as_tensor(a, device=gpu) for a in (X, Y, go))
    k = _lib.VEC_GAUSSIAN if kind == "gaussian" else _lib.VEC_IMQ
    K, dK = ops.vec_kernel_fused(Xg, Yg, k, 1 / h**2, -1 / h**2, grad_out=gog)
    want_sq = VO.pw_dist_sq(X.astype(np.float64), Y.astype(np.float64))
    wantK = np.exp(-0.5 / h**2 * want_sq) if kind == "gaussian" else (1 + 0.5 * want_sq / h**2) ** -0.5
    assert rel(K, wantK) < 1e-5
    want = VO.vec_kernel_weighted_grad(want_sq, X.astype(np.float64), Y.astype(np.float64), go, kind, h, -1 / h**2)
    assert rel(dK, want) < 2e-5
    K2, none = ops.vec_kernel_fused(Xg, Yg, k, 1 / h**2, -1 / h**2, want_grad=False)
    assert none is None and torch.equal(K2, K)
    none, dK2 = ops.vec_kernel_fused(Xg, Yg, k, 1 / h**2, -1 / h**2, grad_out=gog, want_K=False)
    assert none is None and rel(dK2, want) < 2e-5


@pytest.mark.parametrize("A,B,D", [(1024, 1024, 448), (700, 513, 64), (130, 130, 129), (64, 64, 7)])
def test_vec_kernel_fused_is_reproducible_with_its_workspace(gpu, A, B, D):
    """ABI 9: given its workspace the fused kernel stores one partial dK per column split and adds them in split order --
    three calls, another stream busy with GEMMs, return the same bits; the one-launch route (atomics) agrees to rounding"""
    from sigsvgd_amd import _lib, ops

    rng = np.random.default_rng(A + B + D)
    X = torch.as_tensor(rng.normal(size=(A, D)).astype(np.float32), device=gpu)
    Y = torch.as_tensor(rng.normal(size=(B, D)).astype(np.float32), device=gpu)
    h = float(np.sqrt(D))
    side = torch.cuda.Stream()
    junk = torch.randn(1024, 1024, device=gpu)
    outs = []
    for _ in range(3):
        with torch.cuda.stream(side):
            for _ in range(4):
                junk = (junk @ junk).clamp_(-1, 1)
        outs.append(ops.vec_kernel_fused(X, Y, _lib.VEC_GAUSSIAN, 1 / h**2, -1 / h**2))
    torch.cuda.synchronize()
    for K, dK in outs[1:]:
        assert torch.equal(K, outs[0][0]) and torch.equal(dK, outs[0][1])
    Ka, dKa = ops.vec_kernel_fused(X, Y, _lib.VEC_GAUSSIAN, 1 / h**2, -1 / h**2, reproducible=False)
    assert torch.equal(Ka, outs[0][0])
    assert float((dKa - outs[0][1]).abs().max() / outs[0][1].abs().max()) < 2e-6


def test_vec_kernel_fused_metric_and_classes(gpu):
    """Scaled kernels (a non-symmetric metric included) through the fused launch, and the drop-in classes route a
    fixed bandwidth (argument or constant bandwidth_fn) to it with the same results as the two-launch path."""
    from sigsvgd_amd import _lib, ops
    from sigsvgd_amd.kernels import GaussianKernel, ScaledIMQKernel

    rng = np.random.default_rng(11)
    X, Y = rng.normal(size=(90, 33)).astype(np.float32), rng.normal(size=(75, 33)).astype(np.float32)
    R = rng.normal(size=(33, 33))
    for M in ((R @ R.T / 33 + np.eye(33)).astype(np.float32), (R / 6 + np.eye(33)).astype(np.float32)):
        Xg, Yg, Mg = (torch.as_tensor(a, device=gpu) for a in (X, Y, M))
        XM, YM = Xg @ Mg, Yg @ Mg
        K, dK = ops.vec_kernel_fused(Xg, Yg, _lib.VEC_GAUSSIAN, 1 / 3.0**2, -1 / 3.0**2, XM=XM, YM=YM)
        sq = ops.vec_sqdist(Xg, Yg, XM, YM)
        K0, dK0 = ops.vec_kernel(sq, XM, YM, _lib.VEC_GAUSSIAN, 1 / 3.0**2, -1 / 3.0**2)
        assert rel(K, K0.double().cpu().numpy()) < 1e-5 and rel(dK, dK0.double().cpu().numpy()) < 2e-5
    Xs = torch.as_tensor(rng.normal(size=(40, 96)).astype(np.float32), device=gpu)
    two_launch = GaussianKernel()  # median heuristic: distance matrix needed
    K1, d1 = two_launch(Xs, Xs)
    hmed = float(two_launch.get_bandwidth(ops.vec_sqdist(Xs, Xs)))
    K2, d2 = GaussianKernel()(Xs, Xs, h=hmed)                       # fused: h given
    K3, d3 = GaussianKernel(bandwidth_fn=lambda _: hmed)(Xs, Xs)    # fused: constant bandwidth function
    for Kx, dx in ((K2, d2), (K3, d3)):
        assert rel(Kx, K1.double().cpu().numpy()) < 1e-5 and rel(dx, d1.double().cpu().numpy()) < 2e-5
    Ki, di = ScaledIMQKernel()(Xs, Xs, M=torch.eye(96, device=gpu), h=2.0)
    assert Ki.shape == (40, 40) and di.shape == (40, 96) and bool(torch.isfinite(di).all())


@pytest.mark.parametrize("A,B,D,metric", [(2048, 2048, 40, False), (1500, 2300, 72, False), (1100, 2200, 24, True)])
def test_vec_kernel_fused_several_tiles_per_workgroup(gpu, A, B, D, metric):
    """launches whose workgroups walk MORE THAN ONE column tile (more than 512 / row-tiles tiles: the double-buffered stages
    and the shared W tile are reused across tiles) against the two-launch path of vec_kernels.hip on the same inputs"""
    from sigsvgd_amd import _lib, ops

    rng = np.random.default_rng(A + B + D)
    Xg = torch.as_tensor(rng.normal(size=(A, D)).astype(np.float32), device=gpu)
    Yg = torch.as_tensor((rng.normal(size=(B, D)) * 0.9 + 0.1).astype(np.float32), device=gpu)
    h = float(np.sqrt(D))
    XM = YM = None
    if metric:
        R = rng.normal(size=(D, D))
        Mg = torch.as_tensor((R @ R.T / D + np.eye(D)).astype(np.float32), device=gpu)
        XM, YM = Xg @ Mg, Yg @ Mg
    K, dK = ops.vec_kernel_fused(Xg, Yg, _lib.VEC_GAUSSIAN, 1 / h**2, -1 / h**2, XM=XM, YM=YM)
    sq = ops.vec_sqdist(Xg, Yg, XM, YM)
    K0, dK0 = ops.vec_kernel(sq, XM if metric else Xg, YM if metric else Yg, _lib.VEC_GAUSSIAN, 1 / h**2, -1 / h**2)
    assert rel(K, K0.double().cpu().numpy()) < 1e-5 and rel(dK, dK0.double().cpu().numpy()) < 2e-5
    K2, dK2 = ops.vec_kernel_fused(Xg, Yg, _lib.VEC_GAUSSIAN, 1 / h**2, -1 / h**2, XM=XM, YM=YM)
    assert torch.equal(K, K2) and torch.equal(dK, dK2)
