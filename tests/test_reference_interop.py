"""Drop-in check in the build container only (skipped where /root/reference does not exist, e.g. on
the GPU box): the REFERENCE's own SignatureKernel / ScoreEstimator / SVGD / TrajectorySVGD run
unchanged on top of `sigsvgd_amd.sigkernel` registered as `sigkernel` (INTEGRATION.md route A) and
reproduce the committed fixtures.  HIP ops are replaced by the oracle doubles (no GPU here)."""
import os
import sys
import types

import numpy as np
import pytest
import torch

from helpers import golden, patch_ops

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src")), reason="reference tree not present")


@pytest.fixture()
def ref_modules(monkeypatch):
    import sigsvgd_amd.sigkernel as ours

    patch_ops(monkeypatch)
    monkeypatch.setattr(sys, "dont_write_bytecode", True)
    monkeypatch.setitem(sys.modules, "sigkernel", ours)
    monkeypatch.setitem(sys.modules, "signatory", types.ModuleType("signatory"))
    monkeypatch.syspath_prepend(REF)
    for name in [m for m in sys.modules if m == "src" or m.startswith("src.")]:
        monkeypatch.delitem(sys.modules, name)
    import src.inference as inf
    import src.kernels as ker

    torch.autograd.set_detect_anomaly(False)
    yield ker, inf
    for name in [m for m in sys.modules if m == "src" or m.startswith("src.")]:
        sys.modules.pop(name, None)


def _cost_fn(x, w):
    c = w * (x**2).sum((1, 2)) + ((x[:, 1:] - x[:, :-1]) ** 2).sum((1, 2))
    return c, {"aux": c.detach() * 2}


def test_reference_classes_run_on_our_sigkernel(ref_modules):
    ker, inf = ref_modules
    import sigsvgd_amd.sigkernel as ours

    G = golden()
    sk = ker.SignatureKernel(bandwidth_fn=lambda _: 1.5, depth=2)  # reference class, unpatched
    assert isinstance(sk.kernel, ours.SigKernel)
    x = torch.as_tensor(G["c1_X"]).clone().requires_grad_(True)
    K = sk(x, x.detach())
    assert np.allclose(K.detach().numpy(), G["c1_K"], rtol=2e-6)
    est = inf.ScoreEstimator(sk, _cost_fn, {"w": 0.5}, scheduler=None)
    s = inf.SVGD(sk, optimizer_class=torch.optim.Adam, lr=0.05)
    Xp = torch.as_tensor(G["c1_X"]).clone()
    data, _ = s.optimize(Xp, est.score, n_steps=3)
    assert np.abs(data["trace"].numpy() - G["c1_opt_adam_trace"]).max() < 2e-5


def test_reference_trajectory_svgd_isinstance_branch(ref_modules):
    ker, inf = ref_modules
    import sigsvgd_amd.sigkernel as ours

    G = golden()
    actions = torch.as_tensor(G["traj_actions"]).clone().requires_grad_(True)
    Np, hz = actions.shape[0], actions.shape[1]
    pos = torch.cumsum(actions, dim=1)
    traj = torch.cat([torch.zeros(Np, 1, 4), torch.cat([pos, actions], dim=-1)], dim=1)
    sigk = ours.SigKernel(ours.RBFKernel(sigma=(2 + hz) ** 0.5), dyadic_order=1)
    ts = inf.TrajectorySVGD(sigk, gradient_mask=torch.ones(Np, hz, 2), optimizer_class=None, lr=0.1)
    kxx, gk = ts._compute_kernel(actions, trajectories=traj, actions=actions, sample_shape=None)
    assert np.allclose(kxx.numpy(), G["traj_kxx"], rtol=2e-6)
    assert np.abs(gk.numpy() - G["traj_gradk"]).max() / np.abs(G["traj_gradk"]).max() < 1e-5
