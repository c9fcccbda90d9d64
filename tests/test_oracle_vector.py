"""CPU: the vector-kernel / signature oracle against the reference's captured outputs
(tests/golden/ref_vector_kernels.npz) and against independent known answers."""
import math
import os

import numpy as np
import pytest

from oracle import vector_oracle as VO

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_vector_kernels.npz"))
TOL = dict(rtol=1e-12, atol=1e-13)


@pytest.mark.parametrize("name,fn", [("gauss", VO.gaussian), ("imq", VO.imq)])
def test_plain_kernels_match_reference(name, fn):
    X, Y = G["X"], G["Y"]
    K, dK, _ = fn(X, Y, h=0.8)
    np.testing.assert_allclose(K, G[f"{name}_h0.8_K"], **TOL)
    np.testing.assert_allclose(dK, G[f"{name}_h0.8_dK"], **TOL)
    np.testing.assert_allclose(K, G[f"{name}_h0.8_Konly"], **TOL)
    K, dK, _ = fn(X, Y)  # median bandwidth
    np.testing.assert_allclose(K, G[f"{name}_med_K"], rtol=1e-7)  # float32 log inside bw_median
    np.testing.assert_allclose(dK, G[f"{name}_med_dK"], rtol=1e-7, atol=1e-9)
    K, dK, _ = fn(X, X, h=1.1)
    np.testing.assert_allclose(K, G[f"{name}_xx_K"], **TOL)
    np.testing.assert_allclose(dK, G[f"{name}_xx_dK"], **TOL)


@pytest.mark.parametrize("name,fn", [("sgauss", VO.scaled_gaussian), ("simq", VO.scaled_imq)])
def test_scaled_kernels_match_reference(name, fn):
    X, Y, M, Mns = G["X"], G["Y"], G["M"], G["Mns"]
    for key, kw in [("I_h0.8", dict(h=0.8)), ("M_h0.8", dict(M=M, h=0.8)), ("Mns_h1.3", dict(M=Mns, h=1.3))]:
        K, dK, _ = fn(X, Y, **kw)
        np.testing.assert_allclose(K, G[f"{name}_{key}_K"], rtol=1e-11, atol=1e-13)
        np.testing.assert_allclose(dK, G[f"{name}_{key}_dK"], rtol=1e-11, atol=1e-12)
    K, dK, _ = fn(X, Y, M=M)
    np.testing.assert_allclose(K, G[f"{name}_M_med_K"], rtol=1e-7)
    np.testing.assert_allclose(dK, G[f"{name}_M_med_dK"], rtol=1e-7, atol=1e-9)


def test_three_dimensional_particles_are_flattened():
    K, dK, _ = VO.gaussian(G["X3"], G["X3"], h=1.7)
    np.testing.assert_allclose(K, G["gauss_X3_K"], **TOL)
    np.testing.assert_allclose(dK, G["gauss_X3_dK"], **TOL)


def test_weighted_gradient_form_equals_the_dense_sum():
    X, Y, M = G["X"], G["Y"], G["M"]
    sq, _ = VO.scaled_pw_dist_sq(X, Y, M)
    _, dK, _ = VO.scaled_gaussian(X, Y, M=M, h=0.8)
    got = VO.vec_kernel_weighted_grad(sq, X @ M, Y @ M, None, "gaussian", 0.8, -1 / 0.8**2)
    np.testing.assert_allclose(got, dK, rtol=1e-11, atol=1e-13)
    sq = VO.pw_dist_sq(X, Y)
    _, dK, _ = VO.imq(X, Y, h=0.8)
    got = VO.vec_kernel_weighted_grad(sq, X, Y, None, "imq", 0.8, +0.5 / 0.8**2)
    np.testing.assert_allclose(got, dK, rtol=1e-11, atol=1e-13)


# ---- signature -----------------------------------------------------------------------------------------
def test_signature_of_a_straight_line_is_the_tensor_exponential():
    a = np.array([0.7, -0.3, 0.2])
    x = np.linspace(0, 1, 6)[:, None] * a  # starts at 0: basepoint adds a zero increment
    for bp in (True, False):
        S = VO.signature(x[None], 3, basepoint=bp)[0]
        want = np.concatenate([a, np.multiply.outer(a, a).ravel() / 2,
                               np.multiply.outer(np.multiply.outer(a, a), a).ravel() / 6])
        np.testing.assert_allclose(S, want, rtol=1e-13, atol=1e-15)


def test_signature_level2_antisymmetric_part_is_the_levy_area():
    # unit square traversed counter-clockwise from the origin: signed area 1
    x = np.array([[0.0, 0.0], [1.0, 0.0], [1.0, 1.0], [0.0, 1.0], [0.0, 0.0]])
    S = VO.signature(x[None], 2, basepoint=False)[0]
    lvl2 = S[2:].reshape(2, 2)
    np.testing.assert_allclose(S[:2], 0.0, atol=1e-15)
    np.testing.assert_allclose(0.5 * (lvl2[0, 1] - lvl2[1, 0]), 1.0, rtol=1e-14)


@pytest.mark.parametrize("L,C,depth,bp", [(4, 2, 3, True), (5, 3, 3, False), (3, 2, 4, True), (6, 1, 4, True)])
def test_signature_matches_bruteforce_iterated_sums(L, C, depth, bp):
    rng = np.random.default_rng(L * 10 + C)
    x = rng.normal(size=(L, C))
    np.testing.assert_allclose(VO.signature(x[None], depth, bp)[0], VO.signature_bruteforce(x, depth, bp),
                               rtol=1e-12, atol=1e-13)


def test_signature_chen_identity_and_shuffle_product():
    rng = np.random.default_rng(5)
    x = np.cumsum(rng.normal(size=(9, 2)), 0)
    S = VO.signature(x[None], 2, basepoint=False)[0]
    s1, s2 = S[:2], S[2:].reshape(2, 2)
    # shuffle identity at level 2: S^{ab} + S^{ba} = S^a S^b
    np.testing.assert_allclose(s2 + s2.T, np.multiply.outer(s1, s1), rtol=1e-12, atol=1e-13)
    # Chen: concatenating two pieces
    A = VO.signature(x[None, :5], 2, basepoint=False)[0]
    B = VO.signature(x[None, 4:], 2, basepoint=False)[0]
    a1, a2, b1, b2 = A[:2], A[2:].reshape(2, 2), B[:2], B[2:].reshape(2, 2)
    np.testing.assert_allclose(s1, a1 + b1, rtol=1e-12)
    np.testing.assert_allclose(s2, a2 + b2 + np.multiply.outer(a1, b1), rtol=1e-12, atol=1e-13)


def test_signature_channel_count_and_basepoint():
    assert VO.signature_channels(2, 3) == 14 and VO.signature_channels(7, 3) == 399
    x = np.random.default_rng(1).normal(size=(2, 5, 2))
    with_bp = VO.signature(x, 3, True)
    shifted = VO.signature(np.concatenate([np.zeros((2, 1, 2)), x], 1), 3, False)
    np.testing.assert_allclose(with_bp, shifted, rtol=1e-14)
    assert math.isclose(with_bp[0, 0], x[0, -1, 0])  # level 1 = end point - 0


def test_path_sig_kernel_wiring_matches_reference_on_standin_signatory():
    P1, P2 = G["P1"], G["P2"]
    # the reference ignores `h` here: the static GaussianKernel falls back to its median heuristic, whose
    # float32 log(rows + 1) differs by one ulp between torch and numpy for rows + 1 = 7 -> 1e-6 tolerance
    K, dK, _ = VO.path_sig_kernel(P1, P2, depth=3, h=None)
    np.testing.assert_allclose(K, G["psk_d3_h0.9_K"], rtol=1e-6)
    np.testing.assert_allclose(dK, G["psk_d3_h0.9_dK"], rtol=1e-6, atol=1e-8)
    K, dK, _ = VO.path_sig_kernel(P1, P2, depth=2)
    np.testing.assert_allclose(K, G["psk_d2_med_K"], rtol=1e-6)
    K, _, _ = VO.path_sig_kernel(P1, P1, depth=3)
    np.testing.assert_allclose(K, G["psk_d3_Konly"], rtol=1e-6)


@pytest.mark.parametrize("C,depth,bp", [(2, 3, True), (3, 2, False), (2, 4, True), (4, 3, False)])
def test_signature_vjp_is_the_gradient_of_signature(C, depth, bp):
    """the oracle's reverse-mode gradient of the signature: its forward value equals `signature`, and its gradient equals
    central finite differences of `signature` (what the HIP adjoint is compared with on the GPU)"""
    rng = np.random.default_rng(C * 10 + depth)
    X = np.cumsum(0.3 * rng.standard_normal((3, 6, C)), axis=1)
    W = rng.standard_normal((3, VO.signature_channels(C, depth)))
    sig, g = VO.signature_vjp(X, W, depth, bp)
    assert np.abs(sig - VO.signature(X, depth, bp)).max() < 1e-12
    eps = 1e-6
    fd = np.zeros_like(X)
    for idx in np.ndindex(*X.shape):
        Xp, Xm = X.copy(), X.copy()
        Xp[idx] += eps
        Xm[idx] -= eps
        fd[idx] = ((VO.signature(Xp, depth, bp) - VO.signature(Xm, depth, bp)) * W).sum() / (2 * eps)
    assert np.abs(g - fd).max() / np.abs(fd).max() < 1e-7
