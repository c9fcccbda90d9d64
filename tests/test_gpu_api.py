"""GPU tests of the drop-in surface on the real HIP path: the reference-generated fixtures replayed
through sigsvgd_amd's classes on cuda:0, autograd behaviour of SigKernel.compute_Gram, the sharded
partial solve, and size-independent properties at the benchmark size."""
import numpy as np
import pytest
import torch

from helpers import golden
from oracle import sigkernel_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-5
# two fp32-sweep solves of one pair that differ in orientation (the symmetric launch solves (i, j), the ordered one also
# (j, i)) or launch geometry agree to a few ulps PER ENTRY; both are within TOL of the fp64 oracle
SELF = 4e-6


def rel(a, b):
    a = a.detach().double().cpu().numpy() if hasattr(a, "detach") else np.asarray(a, np.float64)
    return float(np.abs(a - b).max() / np.abs(b).max())


def relK(a, b):
    """K parity as north_star states it: max over entries of |K - K_ref| / |K_ref| (K > 0 always)"""
    a = a.detach().double().cpu().numpy() if hasattr(a, "detach") else np.asarray(a, np.float64)
    # (round 4: plain relative error per entry -- rounds 2-3 floored the denominator at 0.1; the 1e-6 only keeps an exact zero
    #  out of it.  Pairs whose K is small against their grid are solved by the exact fp64 pass now: DESIGN.md section 3)
    return float((np.abs(a - b) / np.maximum(np.abs(b), 1e-6)).max())


def _cost_fn(x, w):
    c = w * (x**2).sum((1, 2)) + ((x[:, 1:] - x[:, :-1]) ** 2).sum((1, 2))
    return c, {"aux": c.detach() * 2}


def test_signature_kernel_and_score_estimator_fixtures(gpu):
    from sigsvgd_amd.inference import ScoreEstimator
    from sigsvgd_amd.kernels import SignatureKernel
    from sigsvgd_amd.utils import SquareRootScheduler

    G = golden()
    sk = SignatureKernel(bandwidth_fn=lambda _: 1.5, depth=2)
    x = torch.as_tensor(G["c1_X"], device=gpu).requires_grad_(True)
    K = sk(x, x.detach())
    assert K.dtype == torch.float32 and K.device == x.device
    assert relK(K, G["c1_K"].astype(np.float64)) < TOL
    g = torch.autograd.grad(K.sum(), x)[0]
    assert rel(g, G["c1_gradk"].astype(np.float64)) < TOL
    est = ScoreEstimator(sk, _cost_fn, {"w": 0.5}, scheduler=SquareRootScheduler(1.0), ctx={"device": gpu})
    glp, sd = est.score(x)
    assert rel(glp, G["c1_score_glp"].astype(np.float64)) < TOL
    assert rel(sd["k_xx"], G["c1_score_kxx"].astype(np.float64)) < TOL
    assert rel(sd["grad_k"], G["c1_score_gradk"].astype(np.float64)) < TOL
    _, sd2 = est.score(x)
    assert rel(sd2["grad_k"], G["c1_score_gradk_2nd"].astype(np.float64)) < TOL


@pytest.mark.parametrize("mode", ["adam", "manual"])
def test_svgd_optimize_fixture_on_gpu(gpu, mode):
    """3 SVGD iterations (kernel + gradient + velocity + optimizer) == the reference's loop"""
    from sigsvgd_amd.inference import SVGD, ScoreEstimator
    from sigsvgd_amd.kernels import SignatureKernel

    G = golden()
    sk = SignatureKernel(bandwidth_fn=lambda _: 1.5, depth=2)
    est = ScoreEstimator(sk, _cost_fn, {"w": 0.5}, ctx={"device": gpu})
    s = SVGD(sk, optimizer_class=torch.optim.Adam, lr=0.05) if mode == "adam" else SVGD(sk, optimizer_class=None, lr=0.01)
    Xp = torch.as_tensor(G["c1_X"], device=gpu).clone()
    data, _ = s.optimize(Xp, est.score, n_steps=3)
    ref = G[f"c1_opt_{mode}_trace"].astype(np.float64)
    assert data["trace"].device.type == "cpu" and tuple(data["trace"].shape) == ref.shape
    assert np.abs(data["trace"].numpy() - ref).max() < 3e-5  # Adam normalises the step: absolute tolerance
    assert all(v.device.type == "cpu" for v in data[0].values() if hasattr(v, "device"))
    assert rel(Xp, ref[-1]) < 3e-5
    # device-resident iter_dict variant
    s2 = SVGD(sk, optimizer_class=None, lr=0.01, iter_dict_device=None)
    d2, _ = s2.optimize(torch.as_tensor(G["c1_X"], device=gpu).clone(), est.score, n_steps=1)
    assert d2[0]["k_xx"].device.type == "cuda"


def test_svgd_manual_modes_fixture_on_gpu(gpu):
    from sigsvgd_amd.inference import SVGD, TrajectorySVGD

    class _Dummy:
        pass

    G = golden()

    def fake(x):
        xf = x.detach().flatten(1)
        diff = xf[:, None, :] - xf[None, :, :]
        K = torch.exp(-(diff**2).sum(-1) / 2.0)
        return -x.detach(), {"k_xx": K, "grad_k": (-diff * K[..., None]).sum(1).reshape(x.shape),
                             "loss": (x.detach() ** 2).sum((1, 2))}

    for mode, kw in [("manual", dict(optimizer_class=None, lr=0.1)),
                     ("adagrad", dict(optimizer_class=None, adaptive_gradient=True, lr=0.1)),
                     ("adam", dict(optimizer_class=torch.optim.Adam, lr=0.05))]:
        Xp = torch.as_tensor(G["svgd_X0"], device=gpu).clone()
        data, _ = SVGD(_Dummy(), **kw).optimize(Xp, fake, n_steps=4)
        assert np.abs(data["trace"].numpy() - G[f"svgd_{mode}_trace"]).max() < 2e-5, mode
    ts = TrajectorySVGD(_Dummy(), gradient_mask=torch.as_tensor(G["tsvgd_mask"], device=gpu), optimizer_class=None, lr=0.1)
    v, _ = ts._velocity(torch.as_tensor(G["svgd_X0"], device=gpu), torch.as_tensor(G["svgd_step_in_score"], device=gpu),
                        k_xx=torch.as_tensor(G["svgd_step_in_K"], device=gpu), grad_k=torch.as_tensor(G["svgd_step_in_gk"], device=gpu))
    assert rel(v, G["tsvgd_velocity"].astype(np.float64)) < TOL


def test_fused_adam_matches_reference_state_and_resumes(gpu):
    """The reference's default optimizer (torch.optim.Adam through a closure, svgd.py:20,100-107) runs fused in the
    velocity launch: trace, X.grad, exp_avg / exp_avg_sq equal the fixtures written by the reference's own loop;
    the returned state loads into a fresh torch.optim.Adam; stopping after 2 of 4 steps and resuming from the
    returned state reproduces the uninterrupted run; the kernel really is the fused one (no torch foreach Adam)."""
    from sigsvgd_amd import ops
    from sigsvgd_amd.inference import SVGD

    class _Dummy:
        pass

    G = golden()

    def fake(x):
        xf = x.detach().flatten(1)
        diff = xf[:, None, :] - xf[None, :, :]
        K = torch.exp(-(diff**2).sum(-1) / 2.0)
        return -x.detach(), {"k_xx": K, "grad_k": (-diff * K[..., None]).sum(1).reshape(x.shape),
                             "loss": (x.detach() ** 2).sum((1, 2))}

    calls = {"n": 0}
    orig = ops.svgd_adam

    def counting(*a, **k):
        calls["n"] += 1
        return orig(*a, **k)

    ops.svgd_adam = counting
    try:
        Xp = torch.as_tensor(G["svgd_X0"], device=gpu).clone()
        data, st = SVGD(_Dummy(), optimizer_class=torch.optim.Adam, lr=0.05).optimize(Xp, fake, n_steps=4)
    finally:
        ops.svgd_adam = orig
    assert calls["n"] == 4
    assert np.abs(data["trace"].numpy() - G["svgd_adam_trace"]).max() < 2e-5
    for i in range(4):
        assert np.abs(data[i]["grad"].numpy() - G[f"svgd_adam_grad{i}"]).max() < 2e-5
    s0 = st["state"][0]
    assert float(s0["step"]) == 4.0
    assert rel(s0["exp_avg"], G["svgd_adam_exp_avg"].astype(np.float64)) < 1e-5
    assert rel(s0["exp_avg_sq"], G["svgd_adam_exp_avg_sq"].astype(np.float64)) < 1e-5
    # the state is a regular torch.optim.Adam state_dict
    probe = torch.zeros_like(Xp).requires_grad_(True)
    torch.optim.Adam([probe], lr=0.05).load_state_dict(st)
    # interrupted + resumed == uninterrupted
    Xa = torch.as_tensor(G["svgd_X0"], device=gpu).clone()
    _, st2 = SVGD(_Dummy(), optimizer_class=torch.optim.Adam, lr=0.05).optimize(Xa, fake, n_steps=2)
    SVGD(_Dummy(), optimizer_class=torch.optim.Adam, lr=0.05).optimize(Xa, fake, opt_state=st2, n_steps=2)
    assert rel(Xa, Xp.double().cpu().numpy()) < 1e-6
    # Adam variants the kernel does not implement still go through torch's optimizer
    Xb = torch.as_tensor(G["svgd_X0"], device=gpu).clone()
    SVGD(_Dummy(), optimizer_class=torch.optim.Adam, lr=0.05, amsgrad=True).optimize(Xb, fake, n_steps=1)
    assert bool(torch.isfinite(Xb).all())


@pytest.mark.parametrize("update", ["manual", "adagrad", "adam"])
@pytest.mark.parametrize("shape,dyadic", [((16, 20, 2), 2), ((24, 32, 7), 0), ((16, 128, 14), 0), ((12, 100, 7), 0)])
def test_graphed_iteration_equals_eager(gpu, update, shape, dyadic):
    """the captured HIP graph of one iteration (Gram + gradient + velocity + update) replays to the same
    particles as the eager launches, for the three update rules, on both solvers"""
    from sigsvgd_amd import ops
    from sigsvgd_amd.graph import GraphedSigSVGD
    from sigsvgd_amd.utils.synthetic import synthetic_inputs

    X0, s0 = synthetic_inputs(*shape)
    X0, s0 = X0.to(gpu), s0.to(gpu)
    lr = 1e-2
    if update == "manual":  # a step of ~1e-3 whatever the scale of K (long paths: K ~ 1e10)
        K0, g0 = ops.gram_fwd_bwd(X0, X0, 1.0, dyadic, y_is_x=True)
        lr = 1e-3 / float(ops.svgd_phi(K0, s0, g0).abs().max())
    g = GraphedSigSVGD(X0, inv_h=1.0, dyadic_order=dyadic, lr=lr, update=update)
    g.score.copy_(s0)
    Xe = X0.clone()
    ada = torch.zeros_like(Xe) if update == "adagrad" else None
    adam = ops.AdamState(Xe) if update == "adam" else None
    for it in range(3):
        g.step()
        K, gk = ops.gram_fwd_bwd(Xe, Xe, 1.0, dyadic, y_is_x=True)
        if update == "adam":
            _, Xe = ops.svgd_adam(K, s0, gk, Xe, lr, adam)
        else:
            _, Xe = ops.svgd_phi(K, s0, gk, X=Xe, lr=lr, adagrad_state=ada)
        torch.cuda.synchronize()
        # same kernels, same launch geometry, reductions in a fixed order: the replay is the eager iteration bit for bit
        assert torch.equal(g.K, K) and torch.equal(g.grad_k, gk), (update, it)
        assert torch.equal(g.X, Xe), (update, it)
    assert g.iterations == 3
    if update == "adam":
        assert int(g._adam.step.item()) == 3


def test_route_a_value_equal_buffers_take_the_symmetric_solve(gpu):
    """The reference calls compute_Gram(X.double(), Y.double()) with Y = x.detach(): two buffers, same values
    (src/kernels/_traj_kernels.py:205).  From 32 particles on the wrapper compares them once and launches the
    symmetric variant (each unordered pair once); small or different inputs stay on ordered pairs; results agree."""
    from sigsvgd_amd import ops
    from sigsvgd_amd.sigkernel import RBFKernel, SigKernel
    from sigsvgd_amd.utils.synthetic import synthetic_inputs

    seen = []
    orig_fb, orig_f = ops.gram_fwd_bwd, ops.gram_fwd

    def spy_fb(*a, **k):
        seen.append(bool(a[8] if len(a) > 8 else k.get("y_is_x", False)))
        return orig_fb(*a, **k)

    def spy_f(*a, **k):
        seen.append(bool(k.get("y_is_x", False)))
        return orig_f(*a, **k)

    ops.gram_fwd_bwd, ops.gram_fwd = spy_fb, spy_f
    try:
        sk = SigKernel(RBFKernel(sigma=1.0), dyadic_order=0)
        X, _ = synthetic_inputs(64, 32, 3)
        Xg = X.to(gpu).requires_grad_(True)
        K = sk.compute_Gram(Xg.double(), Xg.detach().double())
        g = torch.autograd.grad(K.sum(), Xg)[0]
        assert seen and all(seen), seen
        seen.clear()
        K_small = sk.compute_Gram(Xg[:8].double(), Xg[:8].detach().double())
        assert seen == [False]
        seen.clear()
        K_diff = sk.compute_Gram(Xg.double(), (Xg.detach() + 1e-3).double())
        assert seen == [False]
    finally:
        ops.gram_fwd_bwd, ops.gram_fwd = orig_fb, orig_f
    Kref, gref = ops.gram_fwd_bwd(Xg.detach(), Xg.detach().clone(), 1.0)  # ordered pairs
    assert relK(K, Kref.double().cpu().numpy()) < SELF and rel(g, gref.double().cpu().numpy()) < TOL
    assert relK(K_small, Kref[:8, :8].double().cpu().numpy()) < SELF and K_diff.shape == (64, 64)


def test_roctx_ranges_can_be_switched_on():
    """SIGSVGD_ROCTX=1: the entry points bracket their launches with roctx ranges (libroctx64 looked up at run time);
    the path must keep working with the hook on -- checked in a fresh process because the switch is read once."""
    import os
    import subprocess
    import sys

    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    code = ("import torch; from sigsvgd_amd import ops; from sigsvgd_amd.utils.synthetic import synthetic_inputs;"
            "X,s=synthetic_inputs(16,32,3); X=X.cuda(); K,g=ops.gram_fwd_bwd(X,X,1.0,y_is_x=True);"
            "v=ops.svgd_phi(K,s.cuda(),g); torch.cuda.synchronize(); print('ok', float(K[0,0]))")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SIGSVGD_ROCTX="1", PYTHONPATH=root),
                       capture_output=True, text=True, timeout=300, cwd=root)
    assert p.returncode == 0 and "ok" in p.stdout, p.stderr[-1500:]


def test_trajectory_svgd_sigkernel_branch_on_gpu(gpu):
    """fp64 upcast / chained autograd through a rollout graph, as DuSt drives it"""
    from sigsvgd_amd.inference import TrajectorySVGD
    from sigsvgd_amd.sigkernel import RBFKernel, SigKernel

    G = golden()
    actions = torch.as_tensor(G["traj_actions"], device=gpu).clone().requires_grad_(True)
    Np, hz = actions.shape[0], actions.shape[1]
    pos = torch.cumsum(actions, dim=1)
    traj = torch.cat([torch.zeros(Np, 1, 4, device=gpu), torch.cat([pos, actions], dim=-1)], dim=1)
    sigk = SigKernel(RBFKernel(sigma=(2 + hz) ** 0.5), dyadic_order=1)
    ts = TrajectorySVGD(sigk, gradient_mask=torch.ones(Np, hz, 2, device=gpu), optimizer_class=None, lr=0.1)
    kxx, gk = ts._compute_kernel(actions, trajectories=traj, actions=actions, sample_shape=None)
    assert rel(kxx, G["traj_kxx"].astype(np.float64)) < TOL and rel(gk, G["traj_gradk"].astype(np.float64)) < TOL
    # the reference's own call pattern: fp64 tensors in, fp64 out
    tau = traj[..., 1:, :2]
    K64 = sigk.compute_Gram(tau.double(), tau.detach().double(), sym=False)
    assert K64.dtype == torch.float64 and relK(K64, G["traj_kxx"].astype(np.float64)) < TOL


def test_compute_gram_autograd_variants(gpu):
    from sigsvgd_amd.sigkernel import LinearKernel, RBFKernel, SigKernel

    rng = np.random.default_rng(0)
    Xn = np.cumsum(0.1 * rng.standard_normal((9, 30, 3)), axis=1).astype(np.float32)
    Yn = np.cumsum(0.1 * rng.standard_normal((7, 30, 3)), axis=1).astype(np.float32)
    for kern, kind, n in [(RBFKernel(0.8), O.RBF, 0), (RBFKernel(0.8), O.RBF, 2), (LinearKernel(), O.LINEAR, 1)]:
        sk = SigKernel(kern, n)
        x = torch.as_tensor(Xn, device=gpu).requires_grad_(True)
        y = torch.as_tensor(Yn, device=gpu)
        K = sk.compute_Gram(x, y)
        Kref, gref = O.gram_backward(Xn, Yn, None, kind, 0.8, n)
        assert relK(K, Kref) < TOL
        (3.0 * K).sum().backward()  # uniform weights: scaled speculative gradient
        assert rel(x.grad, 3.0 * gref) < TOL
        x.grad = None
        w = torch.as_tensor(rng.standard_normal((9, 7)).astype(np.float32), device=gpu)
        K = sk.compute_Gram(x, y)
        (K * w).sum().backward()  # general weights: second fused launch
        assert rel(x.grad, O.gram_backward(Xn, Yn, w.cpu().numpy().astype(np.float64), kind, 0.8, n)[1]) < TOL
    # sym=True (never used by the reference): go + go^T weighting
    sk = SigKernel(RBFKernel(0.8), 0)
    x = torch.as_tensor(Xn, device=gpu).requires_grad_(True)
    K = sk.compute_Gram(x, x.detach(), sym=True)
    w = torch.as_tensor(rng.standard_normal((9, 9)).astype(np.float32), device=gpu)
    (K * w).sum().backward()
    assert rel(x.grad, O.gram_backward(Xn, Xn, w.cpu().numpy().astype(np.float64), O.RBF, 0.8, 0, False, True)[1]) < TOL
    # naive solver + gradient
    skn = SigKernel(RBFKernel(0.8), 1, _naive_solver=True)
    x = torch.as_tensor(Xn, device=gpu).requires_grad_(True)
    skn.compute_Gram(x, y).sum().backward()
    assert rel(x.grad, O.gram_backward(Xn, Yn, None, O.RBF, 0.8, 1, True)[1]) < TOL


def test_default_median_bandwidth_on_gpu(gpu):
    """bandwidth_fn=None -> bw_median of the full distance tensor, as examples/script_planning_robot.py gets"""
    from sigsvgd_amd.kernels import SignatureKernel

    X = np.cumsum(0.3 * np.random.default_rng(1).standard_normal((6, 3, 7)), axis=1).astype(np.float32)
    h = O.bw_median(O.pairwise_sqdist(X, X))
    K = SignatureKernel(depth=3)(torch.as_tensor(X, device=gpu), torch.as_tensor(X, device=gpu))
    assert relK(K, O.gram(X, X, O.RBF, h, 3)) < TOL


@pytest.mark.parametrize("N,T,d,stride", [(24, 64, 7, 2), (40, 32, 3, 3), (20, 20, 14, 4)])
def test_sym_partials_sum_to_full(gpu, N, T, d, stride):
    """multi-GPU building block on one GPU: the per-rank partial solves add up to the full result"""
    from sigsvgd_amd import ops

    X, s = O.synthetic_inputs(N, T, d)
    Xg = X.to(gpu)
    K, g = ops.gram_fwd_bwd(Xg, Xg, 1.0, y_is_x=True)
    parts = [ops.gram_sym_partial(Xg, 1.0, off, stride) for off in range(stride)]
    Ksum = sum(p[0] for p in parts)
    gsum = sum(p[1] for p in parts)
    assert torch.equal(Ksum, K)  # disjoint supports: bitwise
    assert rel(gsum, g.double().cpu().numpy()) < 1e-6
    nz = sum((p[0] != 0).sum().item() for p in parts)
    assert nz == N * N  # every entry owned exactly once (K > 0 everywhere)
    Kref, gref = O.gram_backward(X.numpy(), X.numpy(), None, O.RBF, 1.0, 0)
    assert relK(Ksum, Kref) < TOL and rel(gsum, gref) < TOL


def test_properties_at_benchmark_size(gpu):
    """size-independent checks at N=1024, T=64, d=7 (the oracle covers row samples in test_gpu_fast)"""
    from sigsvgd_amd import ops

    X, score = O.synthetic_inputs(1024, 64, 7)
    Xg, sg = X.to(gpu), score.to(gpu)
    K, g = ops.gram_fwd_bwd(Xg, Xg, 1.0, y_is_x=True)
    # (a) symmetric solve == ordered solve == forward-only solve
    K2, g2 = ops.gram_fwd_bwd(Xg, Xg, 1.0)
    K3 = ops.gram_fwd(Xg, Xg, 1.0)
    assert relK(K2, K.double().cpu().numpy()) < SELF and torch.equal(K2, K3)
    assert rel(g2, g.double().cpu().numpy()) < TOL
    # (b) translation invariance of the RBF signature kernel
    K4, g4 = ops.gram_fwd_bwd(Xg + 3.0, Xg + 3.0, 1.0, y_is_x=True)
    assert relK(K4, K.double().cpu().numpy()) < TOL and rel(g4, g.double().cpu().numpy()) < TOL
    # (c) k(x, constant path) = 1 and boundary: two-point constant paths
    const = Xg[:, :1, :].expand(-1, 64, -1).contiguous()
    assert float((ops.gram_fwd(Xg[:64], const[:64], 1.0) - 1).abs().max()) < 1e-6
    # (d) permutation equivariance: K[perm][:, perm], grad[perm]
    perm = torch.randperm(1024, generator=torch.Generator().manual_seed(0)).to(gpu)
    Kp, gp = ops.gram_fwd_bwd(Xg[perm].contiguous(), Xg[perm].contiguous(), 1.0, y_is_x=True)
    assert relK(Kp, K[perm][:, perm].double().cpu().numpy()) < SELF
    assert rel(gp, g[perm].double().cpu().numpy()) < TOL
    # (e) velocity: linear in (score, grad_k); fused update consistent
    v, Xn = ops.svgd_phi(K, sg, g, X=Xg, lr=1e-3)
    v2 = ops.svgd_phi(K, 2 * sg, 2 * g)
    assert rel(v2, 2 * v.double().cpu().numpy()) < 1e-6
    assert rel(Xn, (Xg - 1e-3 * v).double().cpu().numpy()) < 1e-6
    vref = -((K.double() @ sg.double().flatten(1) - g.double().flatten(1)) / 1024).reshape(v.shape)
    assert rel(v, vref.cpu().numpy()) < TOL


def test_sharded_step_on_rccl_single_rank(gpu):
    """the particle-sharded iteration through torch.distributed's nccl (= RCCL) backend with one
    rank equals the single-GPU iteration (the world-size-2/4 algebra is covered by the gloo tests)"""
    import os

    import torch.distributed as dist

    from sigsvgd_amd import ops
    from sigsvgd_amd.distributed import ShardedSigSVGD

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=gpu)
    try:
        X, s = O.synthetic_inputs(64, 64, 7)
        Xg, sg = X.to(gpu), s.to(gpu)
        sh = ShardedSigSVGD(1.0, 1e-2)
        Xa = sh.step(Xg, sg)
        assert sh.last_gather_grouped is True  # X and score went out as ONE grouped RCCL collective (distributed.py, step 1)
        Kd = sh.gather_gram()
        K, g = ops.gram_fwd_bwd(Xg, Xg, 1.0, y_is_x=True)
        _, Xb = ops.svgd_phi(K, sg, g, X=Xg, lr=1e-2)
        assert rel(Xa, Xb.double().cpu().numpy()) < 1e-6 and torch.equal(Kd, K)
        ref = O.svgd_iteration(X.numpy(), s.numpy(), h=1.0, n=0, lr=1e-2)
        assert rel(Xa, ref["X_new"]) < TOL
        # row-wise fallback (used for shapes outside the symmetric kernel, e.g. T = 128)
        Xc = ShardedSigSVGD(1.0, 1e-2, rowwise=True).step(Xg, sg)
        assert rel(Xc, ref["X_new"]) < TOL
    finally:
        dist.destroy_process_group()


def test_sharded_step_long_paths_on_rccl(gpu):
    """T=128, d=14 (the C5 path shape) through the sharded step under nccl, world size 1: the long-path
    partial solve must equal the single-GPU iteration and the oracle, for the benchmark's smooth paths AND for
    rough paths (scale 0.15: every particle's pair with itself has increments far beyond what a regenerating
    solver tolerates) -- no NaN may reach the particles whichever kernel serves the launch."""
    import os
    import warnings

    import torch.distributed as dist

    from sigsvgd_amd import ops
    from sigsvgd_amd.distributed import ShardedSigSVGD

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = "29547"
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=gpu)
    try:
        for scale, n in ((0.05, 24), (0.15, 12)):
            rng = np.random.default_rng(5)
            X = torch.as_tensor(np.cumsum(scale * rng.standard_normal((n, 128, 14)), axis=1).astype(np.float32))
            s = torch.as_tensor(rng.standard_normal((n, 128, 14)).astype(np.float32))
            Xg, sg = X.to(gpu), s.to(gpu)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore", RuntimeWarning)
                Xa = ShardedSigSVGD(1.0, 1e-3).step(Xg, sg)
            assert bool(torch.isfinite(Xa).all())
            K, g = ops.gram_fwd_bwd(Xg, Xg, 1.0, y_is_x=True)
            assert bool(torch.isfinite(g).all())
            _, Xb = ops.svgd_phi(K, sg, g, X=Xg, lr=1e-3)
            assert rel(Xa, Xb.double().cpu().numpy()) < 1e-6
            ref = O.svgd_iteration(X.numpy(), s.numpy(), h=1.0, n=0, lr=1e-3)
            assert relK(K, ref["K"]) < TOL and rel(g, ref["grad_k"]) < TOL and rel(Xa, ref["X_new"]) < SELF
    finally:
        dist.destroy_process_group()


def test_sigkernel_paired_distance_mmd(gpu):
    """compute_kernel / compute_distance / compute_mmd of the sigkernel surface vs oracle Gram matrices,
    and the gradient of the MMD w.r.t. its first argument vs the oracle's weighted backward."""
    from sigsvgd_amd.sigkernel import RBFKernel, SigKernel

    rng = np.random.default_rng(17)
    X = np.cumsum(0.2 * rng.standard_normal((6, 12, 3)), 1)
    Y = np.cumsum(0.2 * rng.standard_normal((6, 12, 3)), 1)
    sigma, n = 0.8, 1
    h = sigma  # sigkernel's RBFKernel divides by sigma
    Kxx, Kyy, Kxy = (O.gram(a, b, O.RBF, h, n) for a, b in ((X, X), (Y, Y), (X, Y)))
    sk = SigKernel(RBFKernel(sigma), n)
    Xg = torch.as_tensor(X, device=gpu, dtype=torch.float64)
    Yg = torch.as_tensor(Y, device=gpu, dtype=torch.float64)
    assert rel(sk.compute_kernel(Xg, Yg), np.diag(Kxy)) < 1e-6  # increments are kept in fp32
    want_d = np.diag(Kxx).mean() + np.diag(Kyy).mean() - 2 * np.diag(Kxy).mean()
    assert abs(float(sk.compute_distance(Xg, Yg)) - want_d) < 1e-5 * abs(want_d) + 1e-7
    want_m = Kxx.mean() + Kyy.mean() - 2 * Kxy.mean()
    xg = Xg.clone().requires_grad_(True)
    mmd = sk.compute_mmd(xg, Yg)
    assert abs(float(mmd.detach()) - want_m) < 1e-5 * abs(want_m) + 1e-7
    (g,) = torch.autograd.grad(mmd, xg)
    w = np.full((6, 6), 1.0 / 36)
    _, g_xx = O.gram_backward(X, X, w, O.RBF, h, n, False, True)  # sym: both slots of Gram(X, X)
    _, g_xy = O.gram_backward(X, Y, w, O.RBF, h, n)
    assert rel(g, g_xx - 2 * g_xy) < 1e-5


def test_fused_adagrad_step_matches_torch(gpu):
    """sigsvgd_svgd_step: velocity + the reference's simple Adagrad + X - lr*g in one launch, state in place"""
    from sigsvgd_amd import ops

    g = torch.Generator().manual_seed(5)
    N, D = 70, 45
    K = torch.rand(N, N, generator=g)
    s, gk, X = (torch.randn(N, D, generator=g) for _ in range(3))
    mask = (torch.rand(N, D, generator=g) > 0.3).float()
    state = torch.zeros(N, D)
    st_g = state.clone().to(gpu)
    Xc = X.clone()
    Xg = X.to(gpu)
    for _ in range(3):
        v = -((K @ s - gk) / N) * mask
        state = state + v * v
        gsc = v / torch.sqrt(state + 1e-12)
        Xc = Xc - 0.05 * gsc
        vg, Xg = ops.svgd_phi(K.to(gpu), s.to(gpu), gk.to(gpu), mask=mask.to(gpu), X=Xg, lr=0.05, adagrad_state=st_g)
        # entries where K @ s and grad_k nearly cancel carry the fp32 product's rounding at full weight after the
        # normalisation (|g| = 1 on the first step), hence 1e-4 here; X and the state are compared at 1e-5
        assert rel(vg, gsc.double().numpy()) < 1e-4
    assert rel(Xg, Xc.double().numpy()) < 1e-5 and rel(st_g, state.double().numpy()) < 1e-5
    with pytest.raises(ValueError):
        ops.svgd_phi(K.to(gpu), s.to(gpu), gk.to(gpu), adagrad_state=torch.zeros(N, D + 1, device=gpu))


def test_reference_notebook_experiment_statistics(gpu):
    """The only end-to-end signature-kernel numbers the reference stores
    (examples/script_sequential_distribution.ipynb; N=100, T=10, d=2, SignatureKernel(h=5, depth=4), Adam 0.05 x 200):
        cell 12: mean / highest log-probability of the final paths -21.15 / -19.67, "average path length" 3.298
        cell 9 : per-timestep variance 0.05..0.10 at the two ends, 0.44..1.28 inside
    The notebook's run is unseeded on an unknown device and its cell 9 multiplies grad_k by -1 with the remark
    "TODO: Check if this is needed", so this is a statistical pin (3 seeds, bands stated here), not a parity
    fixture; the 5-seed table of both conventions is committed as profiles/r02_notebook_statistics.json:
        sign +1 (the library's own convention, src/inference/score.py:69): -20.87 +- 0.06 / -19.41 / 3.14
            -> reproduces cell 12; variances ~0.25 at every timestep
        sign -1 (cell 9 as written): -27.66 / -25.6 / 5.63; variances 0.25 at the ends, 0.86..1.29 inside
            -> reproduces the SHAPE of cell 9's profile (ends pinned, interior spread, maximum mid-path)
    i.e. the notebook's two stored cells come from runs with different signs."""
    import importlib.util
    import os

    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "sequential_distribution.py")
    spec = importlib.util.spec_from_file_location("sequential_distribution", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    plus = [mod.run(steps=200, seed=s, device=str(gpu), grad_k_sign=+1.0) for s in range(3)]
    mean_lp = np.mean([r["mean_log_prob"] for r in plus])
    max_lp = np.mean([r["max_log_prob"] for r in plus])
    length = np.mean([r["avg_path_length_cell12"] for r in plus])
    assert all(r["moved"] for r in plus)
    assert abs(mean_lp - (-21.15)) < 0.6, mean_lp     # notebook cell 12: -21.15 (measured here -20.87)
    assert abs(max_lp - (-19.67)) < 0.9, max_lp       # notebook cell 12: -19.67 (measured here -19.4)
    assert abs(length - 3.298) < 0.45, length         # notebook cell 12: 3.298  (measured here 3.14)
    minus = [mod.run(steps=200, seed=s, device=str(gpu), grad_k_sign=-1.0) for s in range(3)]
    var = np.mean([r["variance_per_timestep"] for r in minus], axis=0)
    assert np.isfinite(var).all() and var.min() > 1e-2       # no collapse (the notebook's RBF baseline ends at 1e-9)
    assert var[[0, -1]].max() < 0.5 * var[1:-1].min()        # ends pinned more tightly than the interior (cell 9)
    assert 0.4 < var[1:-1].min() and var[1:-1].max() < 1.5   # interior variances in cell 9's range 0.44..1.28
    assert 3 <= int(np.argmax(var)) <= 6                     # maximum mid-path (cell 9: t = 4)


# ---- the planning cost in front of the path (sigsvgd_obstacle_cost) --------------------------------------------
@pytest.mark.parametrize("N,Kx,d,M,Tt,use_splines", [(20, 5, 2, 10, 100, True), (7, 1, 3, 1, 17, True),
                                                     (33, 10, 7, 50, 128, True), (5, 6, 2, 4, 8, False),
                                                     (3, 0, 2, 2, 64, True)])
def test_obstacle_cost_kernel_against_the_oracle(gpu, N, Kx, d, M, Tt, use_splines):
    """HIP cost / trajectories / gradient against the fp64 restatement of script_planning_obstacle_field.py:113-126
    (torch.distributions field + scipy natural spline + autograd)."""
    from oracle import cost_oracle as CO
    from sigsvgd_amd.costs import ObstacleFieldCost

    g = torch.Generator().manual_seed(N * 131 + Kx)
    mean = 0.5 + 4.0 * torch.rand(M, d, generator=g)
    std = 0.2 + 0.5 * torch.rand(M, d, generator=g)
    wts = 0.5 + torch.rand(M, generator=g)
    start, target = 0.5 * torch.rand(d, generator=g), 4.5 + 0.5 * torch.rand(d, generator=g)
    x = torch.linspace(0.5, 4.5, Kx)[None, :, None] + 0.4 * torch.randn(N, Kx, d, generator=g)
    if not use_splines:
        Tt = Kx + 2
    w = (1.3, 0.8)
    cref, tref, gref = CO.cost_and_grad(x, wts, mean, std, start, target, Tt, w, use_splines)
    cost_fn = ObstacleFieldCost(wts.to(gpu), mean.to(gpu), std.to(gpu), start, target, Tt, w, use_splines)
    xg = x.to(gpu).requires_grad_(True)
    cost, aux = cost_fn(xg)
    (gx,) = torch.autograd.grad(-cost.sum(), xg)
    rel = lambda a, b: float(np.abs(a.detach().cpu().numpy() - b).max() / max(np.abs(b).max(), 1e-30))
    assert rel(aux["trajectories"], tref) < 2e-6
    assert rel(cost, cref) < 1e-5
    if Kx:
        assert rel(gx, -gref) < 1e-5
    c2, t2, score = cost_fn.cost_and_score(x.to(gpu))
    assert torch.equal(c2, cost.detach()) and torch.equal(t2, aux["trajectories"])
    if Kx:
        assert torch.equal(score, gx)


def test_obstacle_cost_in_the_score_estimator(gpu):
    """The device cost plugs into ScoreEstimator as the script's cost_fn does (grad_log_p = -d cost / d x)."""
    from oracle import cost_oracle as CO
    from sigsvgd_amd.costs import ObstacleFieldCost
    from sigsvgd_amd.inference import ScoreEstimator

    g = torch.Generator().manual_seed(5)
    mean, std, wts = 0.5 + 4 * torch.rand(8, 2, generator=g), 0.3 * torch.ones(8, 2), torch.ones(8)
    start, target = torch.tensor([0.25, 0.75]), torch.tensor([4.75, 4.5])
    x = (torch.linspace(0.5, 4.5, 5)[None, :, None] + 0.4 * torch.randn(16, 5, 2, generator=g)).to(gpu)
    cost_fn = ObstacleFieldCost(wts.to(gpu), mean.to(gpu), std.to(gpu), start, target)
    est = ScoreEstimator(None, cost_fn, {}, ctx={"device": gpu})
    grad_log_p, dct = est.sgd_score(x.requires_grad_(True))
    _, _, gref = CO.cost_and_grad(x.detach().cpu(), wts, mean, std, start, target)
    assert float(np.abs(grad_log_p.cpu().numpy() + gref).max() / np.abs(gref).max()) < 1e-5
    assert dct["trajectories"].shape == (16, 100, 2)


def test_obstacle_cost_rejects_unsupported_shapes(gpu):
    from sigsvgd_amd import ops

    z = lambda *s: torch.zeros(*s, device=gpu)
    with pytest.raises(RuntimeError, match="unsupported shape"):
        ops.obstacle_cost(z(2, 3, 17), z(17), z(17), z(10, 5), z(1), z(1, 17), 1 + z(1, 17))
    with pytest.raises(RuntimeError, match="gradient output supports"):
        ops.obstacle_cost(z(2, 3, 2), z(2), z(2), z(200, 5), z(1), z(1, 2), 1 + z(1, 2))
    with pytest.raises(ValueError, match="basis must be"):
        ops.obstacle_cost(z(2, 3, 2), z(2), z(2), z(10, 4), z(1), z(1, 2), 1 + z(1, 2))


def test_planning_example_runs_on_the_device_and_lowers_the_cost(gpu):
    import importlib.util
    import os

    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "planning_obstacle_field.py")
    spec = importlib.util.spec_from_file_location("planning_obstacle_field", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = mod.run(steps=60, n_obst=10, seed=0, device=str(gpu))
    assert out["trajectories"].shape == (20, 100, 2) and bool(torch.isfinite(out["trajectories"]).all())
    assert out["cost_final"] < out["cost_initial"]


@pytest.mark.gpu
@pytest.mark.parametrize("A,Tx,B,Ty,d,n", [(9, 20, 7, 13, 3, 0), (5, 40, 6, 64, 7, 0), (4, 100, 5, 70, 2, 0), (6, 10, 5, 7, 2, 4),
                                           (5, 5, 7, 9, 2, 2), (4, 12, 4, 30, 2, 3)])
def test_unequal_path_lengths(gpu, A, Tx, B, Ty, d, n):
    """upstream sigkernel's compute_Gram takes X [A, Tx, d] and Y [B, Ty, d]; here the shorter batch is padded with its last
    point on the host (exact), on whichever kernel the padded shape takes.  Against the numpy oracle's rectangular solve."""
    import numpy as np

    from oracle import sigkernel_oracle as O
    from sigsvgd_amd import ops, sigkernel

    rng = np.random.default_rng(7)
    X = np.cumsum(0.15 * rng.standard_normal((A, Tx, d)), 1).astype(np.float32)
    Y = np.cumsum(0.15 * rng.standard_normal((B, Ty, d)), 1).astype(np.float32)
    go = rng.uniform(0.5, 1.5, (A, B)).astype(np.float32)
    Kref, gref = O.gram_backward(X.astype(np.float64), Y.astype(np.float64), go.astype(np.float64), O.RBF, 1.0, n)
    Xg, Yg, gog = (torch.as_tensor(t, device=gpu) for t in (X, Y, go))
    K, g = ops.gram_fwd_bwd(Xg, Yg, 1.0, n, grad_out=gog)
    Kf = ops.gram_fwd(Xg, Yg, 1.0, n)
    assert tuple(g.shape) == (A, Tx, d)
    relK = lambda a: float((np.abs(a.double().cpu().numpy() - Kref) / np.maximum(np.abs(Kref), 1e-6)).max())
    assert relK(K) < 1e-5 and relK(Kf) < 1e-5
    assert float(np.abs(g.double().cpu().numpy() - gref).max() / np.abs(gref).max()) < 1e-5
    with pytest.raises(ValueError):
        ops.gram_fwd_bwd(Xg, Yg, 1.0, n, y_is_x=True)
    # through the sigkernel-compatible class and autograd
    sk = sigkernel.SigKernel(sigkernel.RBFKernel(sigma=1.0), n)
    Xa = Xg.clone().requires_grad_(True)
    Ka = sk.compute_Gram(Xa, Yg)
    (Ka * gog).sum().backward()
    Kr2, gr2 = Kref, gref  # RBFKernel(sigma): exp(-|x - y|^2 / sigma), sigkernel's convention
    assert float((np.abs(Ka.detach().double().cpu().numpy() - Kr2) / np.maximum(np.abs(Kr2), 1e-6)).max()) < 1e-5
    assert float(np.abs(Xa.grad.double().cpu().numpy() - gr2).max() / np.abs(gr2).max()) < 1e-5
