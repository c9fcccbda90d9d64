"""Workload for `rocprofv3 --kernel-trace --stats`: the reference's own call shape of the signature kernel
(src/kernels/_traj_kernels.py:205: compute_Gram(X.double(), x.detach().double()), two buffers with the same values) at
N = 64 -- the kernel names in the stats show which instantiation served it (…, SYM = true, … = each unordered pair once)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from sigsvgd_amd.sigkernel import RBFKernel, SigKernel
from sigsvgd_amd.utils.synthetic import synthetic_inputs

dev = torch.device("cuda:0")
sk = SigKernel(RBFKernel(sigma=1.0), dyadic_order=0)
X, _ = synthetic_inputs(64, 32, 3)
Xg = X.to(dev).requires_grad_(True)
for _ in range(5):
    K = sk.compute_Gram(Xg.double(), Xg.detach().double())
    (g,) = torch.autograd.grad(K.sum(), Xg)
torch.cuda.synchronize()
print("ok", float(K.sum()), float(g.abs().max()))
