"""time Gram + gradient at C4 with HIP events, a few times (for A/B of reduction-kernel variants under rocprofv3)"""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sigsvgd_amd import ops
from sigsvgd_amd.utils.synthetic import synthetic_inputs
N, T, d = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (1024, 64, 7)
X, _ = synthetic_inputs(N, T, d); X = X.cuda()
for _ in range(12):
    ops.gram_fwd_bwd(X, X, 1.0, 0, y_is_x=True)
torch.cuda.synchronize()
