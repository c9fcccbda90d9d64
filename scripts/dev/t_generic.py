import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sigsvgd_amd.utils.synthetic import synthetic_inputs
from sigsvgd_amd import ops
dev = torch.device('cuda:0')
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time() - t0) / n * 1e3
# default dispatch (refined-grid kernel where its shapes apply) next to the coverage kernel (force_generic)
for (N, T, d, n) in [(16, 20, 2, 2), (64, 20, 2, 2), (256, 20, 2, 2), (30, 5, 2, 5), (100, 5, 2, 5), (50, 3, 7, 6), (1024, 5, 2, 5),
                     (16, 10, 2, 4), (100, 10, 2, 4), (35, 30, 2, 3)]:
    X, s = synthetic_inputs(N, T, d); X = X.to(dev)
    for fg in (False, True):
        print(f"N={N} T={T} d={d} n={n} {'coverage kernel' if fg else 'default dispatch'}: fwd+bwd sym %.3f ms | fwd+bwd ordered %.3f | fwd only %.3f" % (
            t(lambda: ops.gram_fwd_bwd(X, X, 1.0, n, y_is_x=True, force_generic=fg)),
            t(lambda: ops.gram_fwd_bwd(X, X, 1.0, n, force_generic=fg)),
            t(lambda: ops.gram_fwd(X, X, 1.0, n, force_generic=fg))), flush=True)
