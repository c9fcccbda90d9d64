"""C1 / C2 iterations for a rocprofv3 --kernel-trace --stats pass: which launches make up a small iteration"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sigsvgd_amd import ops
from sigsvgd_amd.utils.synthetic import synthetic_inputs
which = sys.argv[1] if len(sys.argv) > 1 else "C2"
N, T, d, n = {"C1": (16, 20, 2, 2), "C2": (128, 32, 7, 0)}[which]
X, s = synthetic_inputs(N, T, d); X = X.cuda(); s = s.cuda()
for _ in range(30):
    K, g = ops.gram_fwd_bwd(X, X, 1.0, n, y_is_x=True)
    ops.svgd_phi(K, s, g, X=X, lr=1e-3)
torch.cuda.synchronize()
