"""The planning-cost oracle (oracle/cost_oracle.py) against what pins it on CPU: finite differences of itself, the
closed forms of the two cost terms, and the product's spline basis (an independent derivation of the same unique
natural spline)."""
import numpy as np
import torch

from oracle import cost_oracle as CO


def _field(M=5, d=2, seed=0):
    g = torch.Generator().manual_seed(seed)
    mean = 0.5 + 4.0 * torch.rand(M, d, generator=g)
    std = 0.3 + 0.4 * torch.rand(M, d, generator=g)
    wts = 0.5 + torch.rand(M, generator=g)
    return wts, mean, std


def test_spline_matrix_matches_the_products_basis():
    from sigsvgd_amd.utils.spline import spline_basis

    for K, Tt in [(2, 7), (3, 10), (7, 100), (12, 33)]:
        B = CO.spline_samples_matrix(K, Tt)
        if K > 2:
            B2 = spline_basis(torch.linspace(0, 1, K).double(), torch.linspace(0, 1, Tt).double())
            assert float((B - B2).abs().max()) < 1e-6  # the product's knot times are fp32 linspace
        assert torch.allclose(B.sum(1), torch.ones(Tt, dtype=torch.float64), atol=1e-12)


def test_cost_terms_closed_form():
    wts, mean, std = _field(3, 2)
    start, target = torch.tensor([0.25, 0.75]), torch.tensor([4.75, 4.5])
    x = torch.stack([start + (target - start) * s for s in (0.25, 0.5, 0.75)])[None]  # knots on the chord
    cost, traj, _ = CO.cost_and_grad(x, wts, mean, std, start, target, timesteps=50, w=(0.0, 2.0))
    assert abs(cost[0] - 2.0 * np.sqrt(((traj[0, 1:] - traj[0, :-1]) ** 2).sum())) < 1e-12
    # uniform samples of a straight line: 49 equal segments
    seg = np.linalg.norm((target - start).double().numpy()) / 49
    assert abs(cost[0] - 2.0 * np.sqrt(49) * seg) < 1e-9
    cost2, traj2, _ = CO.cost_and_grad(x, wts, mean, std, start, target, timesteps=50, w=(3.0, 0.0))
    pi = (wts.double() / wts.double().sum()).numpy()
    z = traj2[0][:, None, :]
    dens = np.exp(-0.5 * ((z - mean.double().numpy()) / std.double().numpy()) ** 2) / (std.double().numpy() * np.sqrt(2 * np.pi))
    assert abs(cost2[0] - 3.0 * (dens.prod(-1) * pi).sum()) < 1e-12


def test_gradient_against_finite_differences():
    wts, mean, std = _field(6, 2, seed=3)
    g = torch.Generator().manual_seed(1)
    start, target = torch.tensor([0.25, 0.75]), torch.tensor([4.75, 4.5])
    x = (2.5 + torch.randn(3, 4, 2, generator=g)).double()
    for use_splines in (True, False):
        _, _, grad = CO.cost_and_grad(x, wts, mean, std, start, target, 40, (1.5, 0.7), use_splines)
        eps = 1e-6
        for idx in [(0, 0, 0), (1, 2, 1), (2, 3, 0)]:
            xp, xm = x.clone(), x.clone()
            xp[idx] += eps
            xm[idx] -= eps
            cp = CO.cost_and_grad(xp, wts, mean, std, start, target, 40, (1.5, 0.7), use_splines)[0].sum()
            cm = CO.cost_and_grad(xm, wts, mean, std, start, target, 40, (1.5, 0.7), use_splines)[0].sum()
            assert abs((cp - cm) / (2 * eps) - grad[idx]) < 1e-6 * max(1.0, abs(grad[idx]))
