"""Kernel trace workload for the small configurations: python3 scripts/dev/c2_trace.py [N T d n]  (default C2)"""
import sys

import torch

sys.path.insert(0, "/root/repo")
from sigsvgd_amd import ops
from sigsvgd_amd.utils.synthetic import synthetic_inputs

N, T, d, n = (int(v) for v in sys.argv[1:5]) if len(sys.argv) >= 5 else (128, 32, 7, 0)
X, s = synthetic_inputs(N, T, d)
X, s = X.cuda(), s.cuda()
for _ in range(50):
    K, g = ops.gram_fwd_bwd(X, X, 1.0, n, y_is_x=True)
    ops.svgd_phi(K, s, g, X=X, lr=1e-3)
torch.cuda.synchronize()
