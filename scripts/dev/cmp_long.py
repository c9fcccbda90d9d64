import sys, time, torch
sys.path.insert(0, '.')
from sigsvgd_amd.utils.synthetic import synthetic_inputs
from sigsvgd_amd import ops
dev = torch.device('cuda:0')
def t(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time() - t0) / n * 1e3
shapes = [(256, 128, 7), (256, 128, 3), (256, 112, 7), (256, 80, 7), (256, 128, 8), (256, 66, 7)]
if len(sys.argv) > 1:
    shapes = [(256, int(t), int(d)) for t in sys.argv[1].split(',') for d in sys.argv[2].split(',')]
for (N, T, d) in shapes:
    X, s = synthetic_inputs(N, T, d); X = X.to(dev)
    r = []
    for sf in (False, True):
        r.append((t(lambda: ops.gram_fwd_bwd(X, X, 1.0, y_is_x=True, stored_forward=sf, check_regime=False)),
                  t(lambda: ops.gram_fwd_bwd(X, X, 1.0, stored_forward=sf, check_regime=False)),
                  t(lambda: ops.gram_fwd(X, X, 1.0, y_is_x=True, stored_forward=sf))))
    print(f"N={N} T={T} d={d}: stream sym {r[0][0]:.2f} ord {r[0][1]:.2f} fwdsym {r[0][2]:.2f} | quad sym {r[1][0]:.2f} ord {r[1][1]:.2f} fwdsym {r[1][2]:.2f}", flush=True)
