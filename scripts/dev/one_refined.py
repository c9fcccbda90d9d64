"""a few launches of the refined-grid and band kernels at the reference's own call shapes, for a --pmc pass
(notebook: 100 x 10 x 2, order 4 -> band kernel; maze: 35 x 30 x 2, order 3 -> band kernel; C1: 16 x 20 x 2, order 2 and the
planning script: 30 x 5 x 2, order 5 -> refined-grid kernel)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sigsvgd_amd import ops
from sigsvgd_amd.utils.synthetic import synthetic_inputs

dev = torch.device("cuda:0")
for (n, t, d, order) in [(100, 10, 2, 4), (35, 30, 2, 3), (16, 20, 2, 2), (30, 5, 2, 5)]:
    X, _ = synthetic_inputs(n, t, d)
    X = X.to(dev)
    for _ in range(2):
        ops.gram_fwd_bwd(X, X, 1.0, order, y_is_x=True)
torch.cuda.synchronize()
