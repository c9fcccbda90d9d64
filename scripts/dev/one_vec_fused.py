"""a few launches of the fused vector kernel (N = 1024, D = 448, Gaussian, fixed bandwidth) for a --pmc pass"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sigsvgd_amd import ops

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
V = torch.randn(1024, 448, generator=g).to(dev)
for _ in range(3):
    ops.vec_kernel_fused(V, V, 0, 1.0 / 900.0, 1.0)  # kind 0: Gaussian
torch.cuda.synchronize()
