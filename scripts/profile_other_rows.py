"""Workload for rocprofv3 --kernel-trace --stats of the kernels bench.py does not touch: the long-path kernel (quadrant
kernel at the C5 path shape N=256 of 4096, T=128, d=14, symmetric, and at N=256, T=100, d=7), the coverage
kernel's successors at C1 (refined-grid kernel) and at the notebook's / the maze script's shapes (band kernel), the vector kernels, the truncated signature, the planning cost and the fused Adam update.  usage: rocprofv3 --kernel-trace --stats ... -- python3 scripts/profile_other_rows.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from sigsvgd_amd.utils.synthetic import synthetic_inputs
from sigsvgd_amd import _lib, ops

dev = torch.device("cuda:0")
X5, s5 = synthetic_inputs(256, 128, 14)
X5, s5 = X5.to(dev), s5.to(dev)
X1, s1 = synthetic_inputs(16, 20, 2)
X1, s1 = X1.to(dev), s1.to(dev)
g = torch.Generator().manual_seed(0)
V = torch.randn(1024, 448, generator=g).to(dev)
P = torch.cumsum(0.3 * torch.randn(1024, 64, 2, generator=g), 1).to(dev)
X7, s7 = synthetic_inputs(256, 100, 7)
X7, s7 = X7.to(dev), s7.to(dev)
Xn = synthetic_inputs(100, 10, 2)[0].to(dev)  # examples/script_sequential_distribution.ipynb: dyadic order 4
Xm = synthetic_inputs(35, 30, 2)[0].to(dev)   # examples/script_control_particle_maze.py: dyadic order 3
from sigsvgd_amd.costs import ObstacleFieldCost
cost_fn = ObstacleFieldCost(torch.ones(10, device=dev), 0.5 + 4 * torch.rand(10, 2, generator=g).to(dev),
                            0.05 * torch.ones(10, 2, device=dev), torch.tensor([0.25, 0.75]), torch.tensor([4.75, 4.5]))
knots = (2.5 + torch.randn(1024, 3, 2, generator=g)).to(dev)
adam = ops.AdamState(X5)
for _ in range(10):
    K, gk = ops.gram_fwd_bwd(X5, X5, 1.0, 0, y_is_x=True, check_regime=False)
    ops.svgd_phi(K, s5, gk, X=X5, lr=1e-3)
    ops.svgd_adam(K, s5, gk, X5.clone(), 0.05, adam)
    K7, gk7 = ops.gram_fwd_bwd(X7, X7, 1.0, 0, y_is_x=True, check_regime=False)
    cost_fn.cost_and_score(knots)
    K, gk = ops.gram_fwd_bwd(X1, X1, 1.0, 2, y_is_x=True)
    ops.svgd_phi(K, s1, gk, X=X1, lr=1e-3)
    ops.gram_fwd_bwd(Xn, Xn, 0.2, 4, y_is_x=True)
    ops.gram_fwd_bwd(Xm, Xm, 1.0 / 32.0, 3, y_is_x=True)
    sq = ops.vec_sqdist(V, V)
    ops.vec_kernel(sq, V, V, _lib.VEC_GAUSSIAN, 1 / 448.0, -1 / 448.0)
    ops.vec_kernel_fused(V, V, _lib.VEC_GAUSSIAN, 1 / 448.0, -1 / 448.0)
    S = ops.signature(P, 3, basepoint=True)
    ops.signature_backward(P, S, 3, basepoint=True)
torch.cuda.synchronize()
print("ok", float(K.sum()), tuple(S.shape))
