import sys, time, torch
sys.path.insert(0, '.')
from sigsvgd_amd.utils.synthetic import synthetic_inputs
from sigsvgd_amd import ops
dev = torch.device('cuda:0')
X, s = synthetic_inputs(256, 128, 14); X = X.to(dev)
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time() - t0) / n * 1e3
print("sym %.3f ms" % t(lambda: ops.gram_fwd_bwd(X, X, 1.0, y_is_x=True, stored_forward=True, check_regime=False)))
