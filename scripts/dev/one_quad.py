import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sigsvgd_amd.utils.synthetic import synthetic_inputs
from sigsvgd_amd import ops
dev = torch.device('cuda:0')
n, t, d = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (256, 128, 14)
X, s = synthetic_inputs(n, t, d); X = X.to(dev)
for _ in range(2):
    ops.gram_fwd_bwd(X, X, 1.0, y_is_x=True, stored_forward=True)
torch.cuda.synchronize()
