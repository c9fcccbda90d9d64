"""Generate tests/golden/ref_fixtures.npz from the REFERENCE's own Python (run in the build container;
/root/reference never travels to the GPU box).

    cd /root/repo && PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

What is captured (SURVEY.md §8c "Golden vectors"):
  * everything on the path that lives in the reference tree, run verbatim: BatchGaussianKernel
    (Gram_matrix / batch_kernel, constant and median bandwidth), bw_median, the three schedulers,
    SVGD.step / SVGD.optimize in its three update modes, TrajectorySVGD._velocity with a mask;
  * the reference's wiring code (SignatureKernel.__call__, ScoreEstimator._pathsig_score,
    TrajectorySVGD._compute_kernel's SigKernel branch, SVGD.optimize on top) executed on top of the
    oracle's `sigkernel`-shaped module, because the real third-party `sigkernel` (setup.py:71) and
    `signatory` are not installed and cannot be fetched offline.  Those fixtures pin dtype casts,
    sign conventions, scheduler scaling and optimizer plumbing -- NOT the PDE arithmetic
    (parity unpinned there, see oracle/sigkernel_oracle.py).

`signatory` is registered as an empty placeholder module only so that `src.kernels` imports
(its PathSigKernel is never called).
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = "/root/reference"
sys.path.insert(0, REF)
sys.dont_write_bytecode = True

import oracle.torch_oracle as oracle_sigkernel  # noqa: E402

sys.modules["sigkernel"] = oracle_sigkernel
sys.modules["signatory"] = types.ModuleType("signatory")

from src.inference import SVGD, ScoreEstimator, TrajectorySVGD  # noqa: E402
from src.kernels import BatchGaussianKernel, SignatureKernel  # noqa: E402
from src.utils.math import bw_median  # noqa: E402
from src.utils.scheduler import CosineScheduler, FactorScheduler, SquareRootScheduler  # noqa: E402

torch.autograd.set_detect_anomaly(False)  # the reference turns it on as an import side effect (mpf.py:9)
out = {}


def npy(t):
    return t.detach().cpu().numpy() if hasattr(t, "detach") else np.asarray(t)


def paths(N, T, d, seed, scale=0.3):
    g = torch.Generator().manual_seed(seed)
    return torch.cumsum(scale * torch.randn(N, T, d, generator=g, dtype=torch.float64), dim=1)


# ---- 1. static kernel ---------------------------------------------------------------------------
X = paths(3, 5, 2, 1)
Y = paths(4, 5, 2, 2)
out["sk_X"], out["sk_Y"] = npy(X), npy(Y)
k = BatchGaussianKernel(bandwidth_fn=lambda _: 0.7)
out["sk_gram_h0.7"] = npy(k.Gram_matrix(X, Y))
out["sk_batch_h0.7"] = npy(k.batch_kernel(X, Y[:3]))
kmed = BatchGaussianKernel()
out["sk_gram_median"] = npy(kmed.Gram_matrix(X, Y))
out["sk_gram_given_h"] = npy(kmed.Gram_matrix(X, Y, h=1.3))

# ---- 2. bw_median -----------------------------------------------------------------------------
g = torch.Generator().manual_seed(3)
sq = torch.rand(5, 4, 6, 6, generator=g, dtype=torch.float64) * 3.0
out["bw_in"] = npy(sq)
out["bw_out"] = npy(bw_median(sq))
out["bw_out_scale2"] = npy(bw_median(sq, bw_scale=2.0))

# ---- 3. schedulers ------------------------------------------------------------------------------
for name, sch in [("sqrt", SquareRootScheduler(2.0)), ("factor", FactorScheduler(1.0, 0.8, 0.3)),
                  ("cosine", CosineScheduler(1.0, 0.1, 8, 2))]:
    out[f"sched_{name}"] = np.array([float(sch()) for _ in range(14)])

# ---- 4. SVGD.step / optimize with a deterministic hand-made score estimator ------------------------
N, T, d = 6, 4, 2
X0 = paths(N, T, d, 5).float()
out["svgd_X0"] = npy(X0)


def fake_estimator(x):
    """standard-normal target, RBF Gram on flattened particles with analytic first-slot gradient"""
    xf = x.detach().flatten(1)
    diff = xf[:, None, :] - xf[None, :, :]
    K = torch.exp(-(diff**2).sum(-1) / 2.0)
    grad_k = (-diff * K[..., None]).sum(1).reshape(x.shape)
    return -x.detach(), {"k_xx": K, "grad_k": grad_k, "loss": (x.detach() ** 2).sum((1, 2))}


class _Dummy:  # SVGD(kernel=...) must be non-None to skip the GaussianKernel default
    pass


for mode, kw in [("manual", dict(optimizer_class=None, lr=0.1)),
                 ("adagrad", dict(optimizer_class=None, adaptive_gradient=True, lr=0.1)),
                 ("adam", dict(optimizer_class=torch.optim.Adam, lr=0.05))]:
    s = SVGD(_Dummy(), **kw)
    Xp = X0.clone()
    data, opt_state = s.optimize(Xp, fake_estimator, n_steps=4)
    out[f"svgd_{mode}_trace"] = npy(data["trace"])
    out[f"svgd_{mode}_final"] = npy(Xp)
    for i in range(4):
        out[f"svgd_{mode}_grad{i}"] = npy(data[i]["grad"])
    out[f"svgd_{mode}_loss3"] = npy(data[3]["loss"])
    if mode == "adam":
        st = opt_state["state"][0]
        out["svgd_adam_exp_avg"] = npy(st["exp_avg"])
        out["svgd_adam_exp_avg_sq"] = npy(st["exp_avg_sq"])

# single step with injected k_xx / grad_k (closed form X - lr * (-(K@s - gk)/N))
s = SVGD(_Dummy(), optimizer_class=None, lr=0.25)
glp, sd = fake_estimator(X0)
Xn, it = s.step(X0, glp, None, **sd)
out["svgd_step_in_score"] = npy(glp)
out["svgd_step_in_K"] = npy(sd["k_xx"])
out["svgd_step_in_gk"] = npy(sd["grad_k"])
out["svgd_step_out_X"] = npy(Xn)
out["svgd_step_out_grad"] = npy(it["grad"])

# ---- 5. TrajectorySVGD._velocity with a gradient mask ----------------------------------------------
mask = torch.ones(N, T, d)
mask[:, 0, :] = 0.0
ts = TrajectorySVGD(_Dummy(), gradient_mask=mask, optimizer_class=None, lr=0.1)
v, _ = ts._velocity(X0, glp, **sd)
out["tsvgd_mask"] = npy(mask)
out["tsvgd_velocity"] = npy(v)

# ---- 6. reference wiring on top of the oracle's sigkernel: C1-sized (N=16, T=20, d=2, depth 2) ------
Xc = paths(16, 20, 2, 7, scale=0.1).float()
out["c1_X"] = npy(Xc)
sk = SignatureKernel(bandwidth_fn=lambda _: 1.5, depth=2)
xr = Xc.clone().requires_grad_(True)
K = sk(xr, xr.detach())
out["c1_K"] = npy(K)  # fp32 after the reference's fp64 upcast / cast-back
out["c1_gradk"] = npy(torch.autograd.grad(K.sum(), xr)[0])


def cost_fn(x, w):
    """toy differentiable cost: squared distance to the origin + path length"""
    c = w * (x**2).sum((1, 2)) + ((x[:, 1:] - x[:, :-1]) ** 2).sum((1, 2))
    return c, {"aux": c.detach() * 2}


est = ScoreEstimator(sk, cost_fn, {"w": 0.5}, scheduler=SquareRootScheduler(1.0))
xr = Xc.clone().requires_grad_(True)
glp, sd = est.score(xr)
out["c1_score_glp"] = npy(glp)
out["c1_score_kxx"] = npy(sd["k_xx"])
out["c1_score_gradk"] = npy(sd["grad_k"])
glp2, sd2 = est.score(xr)  # second call: scheduler value 1/sqrt(2)
out["c1_score_gradk_2nd"] = npy(sd2["grad_k"])

est = ScoreEstimator(sk, cost_fn, {"w": 0.5}, scheduler=None)
s = SVGD(sk, optimizer_class=torch.optim.Adam, lr=0.05)
Xp = Xc.clone()
data, _ = s.optimize(Xp, est.score, n_steps=3)
out["c1_opt_adam_trace"] = npy(data["trace"])
out["c1_opt_adam_keys"] = np.array(sorted(data[0].keys()))
s = SVGD(sk, optimizer_class=None, lr=0.01)
Xp = Xc.clone()
data, _ = s.optimize(Xp, est.score, n_steps=3)
out["c1_opt_manual_trace"] = npy(data["trace"])

# ---- 7. TrajectorySVGD SigKernel branch: k_xx on rolled-out positions, grad chained to actions -------
Np, hz = 8, 10
g = torch.Generator().manual_seed(11)
actions = (0.2 * torch.randn(Np, hz, 2, generator=g)).requires_grad_(True)
state0 = torch.zeros(Np, 1, 4)
pos = torch.cumsum(actions, dim=1)  # toy rollout: integrate actions
traj = torch.cat([state0, torch.cat([pos, actions], dim=-1)], dim=1)  # [Np, hz+1, 4]
sigk = oracle_sigkernel.SigKernel(oracle_sigkernel.RBFKernel(sigma=(2 + hz) ** 0.5), dyadic_order=1)
ts = TrajectorySVGD(sigk, gradient_mask=torch.ones(Np, hz, 2), optimizer_class=None, lr=0.1)
kxx, gk = ts._compute_kernel(actions, trajectories=traj, actions=actions, sample_shape=None)
out["traj_actions"] = npy(actions)
out["traj_kxx"] = npy(kxx)
out["traj_gradk"] = npy(gk)

path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_fixtures.npz")
np.savez_compressed(path, **out)
print("wrote", path, "with", len(out), "arrays,", os.path.getsize(path), "bytes")
