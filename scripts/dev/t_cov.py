import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sigsvgd_amd.utils.synthetic import synthetic_inputs
from sigsvgd_amd import ops
dev = torch.device('cuda:0')
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time() - t0) / n * 1e3
for (N, T, d, n) in [(100, 10, 2, 4), (35, 30, 2, 3), (256, 20, 2, 2), (16, 10, 2, 4)]:
    X, s = synthetic_inputs(N, T, d); X = X.to(dev)
    print(f"N={N} T={T} d={d} n={n} coverage: sym %.3f ms | ordered %.3f | fwd %.3f" % (
        t(lambda: ops.gram_fwd_bwd(X, X, 1.0, n, y_is_x=True, force_generic=True)),
        t(lambda: ops.gram_fwd_bwd(X, X, 1.0, n, force_generic=True)),
        t(lambda: ops.gram_fwd(X, X, 1.0, n, force_generic=True))), flush=True)
