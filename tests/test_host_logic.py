"""CPU tests of the host side (the Python mirror of the reference's operator API): every fixture in
tests/golden/ref_fixtures.npz was produced by the REFERENCE's code (tests/golden/make_golden.py);
here the same inputs go through sigsvgd_amd's classes.  The HIP ops are replaced by oracle-backed
doubles (tests/helpers.py) -- the product itself has no CPU path, which is also asserted."""
import numpy as np
import pytest
import torch

from helpers import golden, patch_ops

ATOL = 2e-6


def close(a, b, rtol=2e-5, atol=ATOL):
    a = a.detach().cpu().numpy() if hasattr(a, "detach") else np.asarray(a)
    return np.allclose(a, b, rtol=rtol, atol=atol)


# ---- no CPU fallback ---------------------------------------------------------------------------------
def test_product_path_rejects_cpu_tensors():
    from sigsvgd_amd import ops
    from sigsvgd_amd.kernels import SignatureKernel

    x = torch.randn(3, 5, 2)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.gram_fwd(x, x, 1.0)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.gram_fwd_bwd(x, x, 1.0)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.svgd_phi(torch.eye(3), x, x)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        SignatureKernel(lambda _: 1.0, depth=1)(x, x)


def test_missing_extension_fails_loudly(monkeypatch, tmp_path):
    from sigsvgd_amd import _lib

    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="not built"):
        _lib.load()


# ---- static kernel, bandwidth, schedulers --------------------------------------------------------------
def test_batch_gaussian_kernel_fixture():
    from sigsvgd_amd.kernels import BatchGaussianKernel

    G = golden()
    X, Y = torch.as_tensor(G["sk_X"]), torch.as_tensor(G["sk_Y"])
    k = BatchGaussianKernel(bandwidth_fn=lambda _: 0.7)
    assert close(k.Gram_matrix(X, Y), G["sk_gram_h0.7"], 1e-13, 0)
    assert close(k.batch_kernel(X, Y[:3]), G["sk_batch_h0.7"], 1e-13, 0)
    assert close(k(X, Y[:3]), G["sk_batch_h0.7"], 1e-13, 0)
    km = BatchGaussianKernel()
    assert close(km.Gram_matrix(X, Y), G["sk_gram_median"], 1e-13, 0)
    assert close(km.Gram_matrix(X, Y, h=1.3), G["sk_gram_given_h"], 1e-13, 0)
    # the fused path resolves the same bandwidths without / with the distance tensor
    assert k.inv_bandwidth(X, Y) == pytest.approx(1 / 0.7)
    from oracle import sigkernel_oracle as O

    assert km.inv_bandwidth(X, Y) == pytest.approx(1 / O.bw_median(O.pairwise_sqdist(G["sk_X"], G["sk_Y"])), rel=1e-12)
    with pytest.raises(ValueError):
        BatchGaussianKernel(bandwidth_fn=3.0)


def test_bw_median_fixture():
    from sigsvgd_amd.utils import bw_median

    G = golden()
    assert float(bw_median(torch.as_tensor(G["bw_in"]))) == pytest.approx(float(G["bw_out"]), rel=1e-14)
    assert float(bw_median(torch.as_tensor(G["bw_in"]), bw_scale=2.0)) == pytest.approx(float(G["bw_out_scale2"]), rel=1e-14)


def test_schedulers_fixture():
    from sigsvgd_amd.utils import CosineScheduler, FactorScheduler, SquareRootScheduler

    G = golden()
    for name, sch in [("sqrt", SquareRootScheduler(2.0)), ("factor", FactorScheduler(1.0, 0.8, 0.3)),
                      ("cosine", CosineScheduler(1.0, 0.1, 8, 2))]:
        vals = np.array([float(sch()) for _ in range(14)])
        assert np.allclose(vals, G[f"sched_{name}"], rtol=1e-6), name
    s = SquareRootScheduler(2.0)
    assert float(s(update_epoch=False)) == float(s(update_epoch=False))


# ---- SVGD.step / optimize ----------------------------------------------------------------------------------
def _fake_estimator(x):
    xf = x.detach().flatten(1)
    diff = xf[:, None, :] - xf[None, :, :]
    K = torch.exp(-(diff**2).sum(-1) / 2.0)
    grad_k = (-diff * K[..., None]).sum(1).reshape(x.shape)
    return -x.detach(), {"k_xx": K, "grad_k": grad_k, "loss": (x.detach() ** 2).sum((1, 2))}


class _Dummy:
    pass


@pytest.mark.parametrize("mode,kw", [("manual", dict(optimizer_class=None, lr=0.1)),
                                     ("adagrad", dict(optimizer_class=None, adaptive_gradient=True, lr=0.1)),
                                     ("adam", dict(optimizer_class=torch.optim.Adam, lr=0.05))])
def test_svgd_optimize_modes_fixture(monkeypatch, mode, kw):
    from sigsvgd_amd.inference import SVGD

    patch_ops(monkeypatch)
    G = golden()
    s = SVGD(_Dummy(), **kw)
    Xp = torch.as_tensor(G["svgd_X0"]).clone()
    data, opt_state = s.optimize(Xp, _fake_estimator, n_steps=4)
    assert close(data["trace"], G[f"svgd_{mode}_trace"])
    assert close(Xp, G[f"svgd_{mode}_final"])  # written back in place
    for i in range(4):
        assert close(data[i]["grad"], G[f"svgd_{mode}_grad{i}"]), i
        assert set(data[i].keys()) == {"k_xx", "grad_k", "loss", "grad"}
        assert all(v.device.type == "cpu" for v in data[i].values() if hasattr(v, "device"))
    assert close(data[3]["loss"], G[f"svgd_{mode}_loss3"])
    if mode == "adam":
        st = opt_state["state"][0]
        assert close(st["exp_avg"], G["svgd_adam_exp_avg"]) and close(st["exp_avg_sq"], G["svgd_adam_exp_avg_sq"])
    else:
        assert opt_state is None


def test_svgd_step_injected_kernel_fixture(monkeypatch):
    from sigsvgd_amd.inference import SVGD

    patch_ops(monkeypatch)
    G = golden()
    s = SVGD(_Dummy(), optimizer_class=None, lr=0.25)
    X0 = torch.as_tensor(G["svgd_X0"])
    Xn, it = s.step(X0, torch.as_tensor(G["svgd_step_in_score"]), None, k_xx=torch.as_tensor(G["svgd_step_in_K"]),
                    grad_k=torch.as_tensor(G["svgd_step_in_gk"]))
    assert close(Xn, G["svgd_step_out_X"]) and close(it["grad"], G["svgd_step_out_grad"])


def test_svgd_errors():
    from sigsvgd_amd.inference import SVGD

    from sigsvgd_amd.kernels import GaussianKernel

    assert isinstance(SVGD(None).kernel, GaussianKernel)  # reference default, svgd.py:24-25
    with pytest.raises(ValueError):
        SVGD(_Dummy(), optimizer_class=None, lr=0.1)._velocity(torch.zeros(2, 3, 1), None)


def test_svgd_opt_state_roundtrip(monkeypatch):
    """optimize(opt_state=...) resumes Adam exactly (reference svgd.py:130-133,158)"""
    from sigsvgd_amd.inference import SVGD

    patch_ops(monkeypatch)
    G = golden()
    s = SVGD(_Dummy(), optimizer_class=torch.optim.Adam, lr=0.05)
    Xa = torch.as_tensor(G["svgd_X0"]).clone()
    d1, st = s.optimize(Xa, _fake_estimator, n_steps=2)
    d2, _ = s.optimize(Xa, _fake_estimator, opt_state=st, n_steps=2)
    assert close(Xa, G["svgd_adam_final"])


def test_trajectory_svgd_mask_fixture(monkeypatch):
    from sigsvgd_amd.inference import TrajectorySVGD

    patch_ops(monkeypatch)
    G = golden()
    ts = TrajectorySVGD(_Dummy(), gradient_mask=torch.as_tensor(G["tsvgd_mask"]), optimizer_class=None, lr=0.1)
    v, _ = ts._velocity(torch.as_tensor(G["svgd_X0"]), torch.as_tensor(G["svgd_step_in_score"]),
                        k_xx=torch.as_tensor(G["svgd_step_in_K"]), grad_k=torch.as_tensor(G["svgd_step_in_gk"]))
    assert close(v, G["tsvgd_velocity"])


# ---- signature-kernel wiring (C1-sized fixtures captured through the reference's own classes) -----------
def _cost_fn(x, w):
    c = w * (x**2).sum((1, 2)) + ((x[:, 1:] - x[:, :-1]) ** 2).sum((1, 2))
    return c, {"aux": c.detach() * 2}


def test_signature_kernel_call_and_autograd_fixture(monkeypatch):
    from sigsvgd_amd.kernels import SignatureKernel

    patch_ops(monkeypatch)
    G = golden()
    sk = SignatureKernel(bandwidth_fn=lambda _: 1.5, depth=2, bandwidth=123.0)  # unknown kwarg is swallowed
    x = torch.as_tensor(G["c1_X"]).clone().requires_grad_(True)
    K = sk(x, x.detach())
    assert K.dtype == torch.float32 and close(K, G["c1_K"])
    g = torch.autograd.grad(K.sum(), x)[0]
    assert close(g, G["c1_gradk"], atol=1e-5)
    # general grad_output goes through the second fused launch
    K2 = sk(x, x.detach())
    w = torch.linspace(0.5, 1.5, K2.numel()).reshape(K2.shape)
    g2 = torch.autograd.grad((K2 * w).sum(), x)[0]
    from oracle import sigkernel_oracle as O

    ref = O.gram_backward(G["c1_X"], G["c1_X"], w.numpy(), O.RBF, 1.5, 2)[1]
    assert close(g2, ref, atol=1e-5)
    # gradient only for the first argument
    y = torch.as_tensor(G["c1_X"]).clone().requires_grad_(True)
    K3 = sk(x.detach(), y)
    assert torch.autograd.grad(K3.sum(), y, allow_unused=True)[0] is None  # like upstream: None for Y


def test_score_estimator_fixture(monkeypatch):
    from sigsvgd_amd.inference import ScoreEstimator
    from sigsvgd_amd.kernels import SignatureKernel
    from sigsvgd_amd.utils import SquareRootScheduler

    patch_ops(monkeypatch)
    G = golden()
    sk = SignatureKernel(bandwidth_fn=lambda _: 1.5, depth=2)
    est = ScoreEstimator(sk, _cost_fn, {"w": 0.5}, scheduler=SquareRootScheduler(1.0))
    assert est.score == est._pathsig_score
    x = torch.as_tensor(G["c1_X"]).clone().requires_grad_(True)
    glp, sd = est.score(x)
    assert close(glp, G["c1_score_glp"]) and close(sd["k_xx"], G["c1_score_kxx"])
    assert close(sd["grad_k"], G["c1_score_gradk"], atol=1e-5)
    assert set(sd.keys()) == {"k_xx", "grad_k", "loss", "aux"}
    _, sd2 = est.score(x)
    assert close(sd2["grad_k"], G["c1_score_gradk_2nd"], atol=1e-5)


@pytest.mark.parametrize("mode", ["adam", "manual"])
def test_svgd_optimize_with_signature_kernel_fixture(monkeypatch, mode):
    from sigsvgd_amd.inference import SVGD, ScoreEstimator
    from sigsvgd_amd.kernels import SignatureKernel

    patch_ops(monkeypatch)
    G = golden()
    sk = SignatureKernel(bandwidth_fn=lambda _: 1.5, depth=2)
    est = ScoreEstimator(sk, _cost_fn, {"w": 0.5}, scheduler=None)
    s = SVGD(sk, optimizer_class=torch.optim.Adam, lr=0.05) if mode == "adam" else SVGD(sk, optimizer_class=None, lr=0.01)
    Xp = torch.as_tensor(G["c1_X"]).clone()
    data, _ = s.optimize(Xp, est.score, n_steps=3)
    assert close(data["trace"], G[f"c1_opt_{mode}_trace"], atol=2e-5)
    if mode == "adam":
        assert sorted(data[0].keys()) == list(G["c1_opt_adam_keys"])


def test_svgd_compute_kernel_fallback(monkeypatch):
    """without injected k_xx/grad_k SVGD asks the kernel itself (reference svgd.py:36-44)"""
    from sigsvgd_amd.inference import SVGD
    from sigsvgd_amd.kernels import SignatureKernel

    patch_ops(monkeypatch)
    G = golden()
    sk = SignatureKernel(bandwidth_fn=lambda _: 1.5, depth=2)
    s = SVGD(sk, optimizer_class=None, lr=0.1)
    X = torch.as_tensor(G["c1_X"])
    K, gk = s._compute_kernel(X)
    assert close(K, G["c1_K"]) and close(gk.reshape(X.shape), G["c1_gradk"], atol=1e-5)


def test_trajectory_svgd_sigkernel_branch_fixture(monkeypatch):
    from sigsvgd_amd.inference import TrajectorySVGD
    from sigsvgd_amd.sigkernel import RBFKernel, SigKernel

    patch_ops(monkeypatch)
    G = golden()
    actions = torch.as_tensor(G["traj_actions"]).clone().requires_grad_(True)
    Np, hz = actions.shape[0], actions.shape[1]
    pos = torch.cumsum(actions, dim=1)
    traj = torch.cat([torch.zeros(Np, 1, 4), torch.cat([pos, actions], dim=-1)], dim=1)
    sigk = SigKernel(RBFKernel(sigma=(2 + hz) ** 0.5), dyadic_order=1)
    ts = TrajectorySVGD(sigk, gradient_mask=torch.ones(Np, hz, 2), optimizer_class=None, lr=0.1)
    kxx, gk = ts._compute_kernel(actions, trajectories=traj, actions=actions, sample_shape=None)
    assert close(kxx, G["traj_kxx"]) and close(gk, G["traj_gradk"], atol=1e-5)


def test_sigkernel_module_surface():
    import sigsvgd_amd.sigkernel as sk

    k = sk.SigKernel(sk.RBFKernel(0.5), 3)
    assert k.dyadic_order == 3 and k._naive_solver is False and hasattr(k, "compute_Gram")
    X = torch.randn(2, 4, 3, dtype=torch.float64)
    from oracle import sigkernel_oracle as O

    assert np.allclose(sk.RBFKernel(0.5).Gram_matrix(X, X).numpy(), O.static_gram(X.numpy(), X.numpy(), O.RBF, 0.5))
    assert np.allclose(sk.LinearKernel().batch_kernel(X, X).numpy(), O.static_batch(X.numpy(), X.numpy(), O.LINEAR))
    with pytest.raises(NotImplementedError):
        sk.SigKernel(object(), 1).compute_Gram(X, X)


def test_package_synthetic_inputs_match_the_oracle_generator():
    """bench.py takes its inputs from the package (no dependency on oracle/); both generators must agree bit for bit"""
    import torch

    from oracle import sigkernel_oracle as O
    from sigsvgd_amd.utils.synthetic import synthetic_inputs

    for shape in [(16, 20, 2), (8, 64, 7)]:
        Xa, sa = synthetic_inputs(*shape)
        Xb, sb = O.synthetic_inputs(*shape)
        assert torch.equal(Xa, Xb) and torch.equal(sa, sb)


def test_constant_bandwidth_detection():
    """`_VectorKernel._constant_bandwidth`: a bandwidth function that ignores the distances is recognised once per
    kernel object (the fused one-launch path then applies); the median heuristic and data-dependent functions are not."""
    from sigsvgd_amd.kernels import GaussianKernel, IMQKernel

    assert GaussianKernel()._constant_bandwidth() is None                      # bw_median: data-dependent
    assert GaussianKernel(bandwidth_fn=lambda _: 0.2)._constant_bandwidth() == 0.2
    assert IMQKernel(bandwidth_fn=lambda sq: sq.mean())._constant_bandwidth() is None
    calls = []

    def needs_matrix(sq):  # a function that only works on a real [N, N] matrix
        calls.append(1)
        return sq[0, 1]

    k = GaussianKernel(bandwidth_fn=needs_matrix)
    assert k._constant_bandwidth() is None and k._constant_bandwidth() is None
    assert len(calls) == 1  # probed once (the first probe raised), then cached
    assert GaussianKernel(bandwidth_fn=lambda _: -1.0)._constant_bandwidth() is None  # not a usable bandwidth
    # functions that depend on the data or the shape but would agree on two one-element probes are NOT frozen
    # (they used to be): any touch of the argument means data-dependent
    for fn in (lambda sq: sq.median().clamp(min=3.0), lambda sq: max(float(sq.median()), 5.0),
               lambda sq: sq.shape[0] ** -0.2, lambda sq: 0.3 if sq.numel() < 10 else 0.7):
        assert GaussianKernel(bandwidth_fn=fn)._constant_bandwidth() is None
    # reassigning get_bandwidth re-probes instead of keeping the old constant
    k2 = GaussianKernel(bandwidth_fn=lambda _: 0.2)
    assert k2._constant_bandwidth() == 0.2
    k2.get_bandwidth = lambda _: 0.5
    assert k2._constant_bandwidth() == 0.5
    k2.get_bandwidth = lambda sq: sq.mean()
    assert k2._constant_bandwidth() is None


def test_unequal_lengths_by_padding_is_exact():
    """ops pads the shorter batch of paths with its last point and folds the gradient of the copies back: zero increments copy
    the PDE solution along the added rows, so K and the first-slot gradient equal those of the unpadded problem (numpy oracle,
    which solves rectangular grids directly; reference: upstream sigkernel accepts unequal lengths, no caller uses them)"""
    import numpy as np
    import torch

    from oracle import sigkernel_oracle as O
    from sigsvgd_amd import ops

    rng = np.random.default_rng(0)
    for (A, Tx, B, Ty, d, n) in [(3, 7, 4, 5, 2, 1), (2, 4, 3, 9, 3, 0), (3, 6, 2, 6, 2, 2)]:
        X = np.cumsum(0.2 * rng.standard_normal((A, Tx, d)), 1)
        Y = np.cumsum(0.2 * rng.standard_normal((B, Ty, d)), 1)
        go = rng.uniform(0.5, 1.5, (A, B))
        K, g = O.gram_backward(X, Y, go, O.RBF, 1.3, n)
        T = max(Tx, Ty)
        Xp = ops.pad_to_length(torch.as_tensor(X), T).numpy()
        Yp = ops.pad_to_length(torch.as_tensor(Y), T).numpy()
        assert Xp.shape == (A, T, d) and Yp.shape == (B, T, d)
        Kp, gp = O.gram_backward(Xp, Yp, go, O.RBF, 1.3, n)
        gf = ops.fold_padded_grad(torch.as_tensor(gp), Tx).numpy()
        assert np.abs(K - Kp).max() < 1e-13
        assert gf.shape == g.shape and np.abs(g - gf).max() < 1e-13
