import sys, time, torch
sys.path.insert(0, '.')
from sigsvgd_amd.utils.synthetic import synthetic_inputs
from sigsvgd_amd import ops
dev = torch.device('cuda:0')
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time() - t0) / n * 1e3
for (N, T, d, n) in [(16, 20, 2, 2), (16, 20, 2, 0), (64, 20, 2, 2), (256, 20, 2, 2), (16, 10, 2, 4), (100, 10, 2, 4)]:
    X, s = synthetic_inputs(N, T, d); X = X.to(dev)
    print(f"N={N} T={T} d={d} n={n}: fwd+bwd sym %.3f ms | fwd+bwd ordered %.3f | fwd only %.3f" % (
        t(lambda: ops.gram_fwd_bwd(X, X, 1.0, n, y_is_x=True, force_generic=True)),
        t(lambda: ops.gram_fwd_bwd(X, X, 1.0, n, force_generic=True)),
        t(lambda: ops.gram_fwd(X, X, 1.0, n, force_generic=True))), flush=True)
