"""a few launches of the band and refined-grid kernels at the reference's own call shapes, for a --pmc pass
(notebook: 100 x 10 x 2, order 4; maze: 35 x 30 x 2, order 3; C1: 16 x 20 x 2, order 2; the planning script: 30 x 5 x 2, order 5
-> the band kernel's band-parallel schedule; 150 x 10 x 2, order 4 -> its serial schedule; 150 x 5 x 2, order 5 -> the
refined-grid kernel of gram_dyad.hip, which keeps the large launches of 64 .. 128 cells)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sigsvgd_amd import ops
from sigsvgd_amd.utils.synthetic import synthetic_inputs

dev = torch.device("cuda:0")
for (n, t, d, order) in [(100, 10, 2, 4), (35, 30, 2, 3), (16, 20, 2, 2), (30, 5, 2, 5), (150, 10, 2, 4), (150, 5, 2, 5)]:
    X, _ = synthetic_inputs(n, t, d)
    X = X.to(dev)
    for _ in range(2):
        ops.gram_fwd_bwd(X, X, 1.0, order, y_is_x=True)
torch.cuda.synchronize()
