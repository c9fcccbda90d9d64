"""One rank's share of the sharded C4 step on ONE GPU (no collectives): partial solve + velocity for every rank of
G = 1, 2, 4, 8 with folded and cyclic tile ownership; worst rank / mean tells the balance."""
import sys
import time

import torch

sys.path.insert(0, ".")
from sigsvgd_amd import ops  # noqa: E402
from sigsvgd_amd.utils.synthetic import synthetic_inputs  # noqa: E402

dev = torch.device("cuda:0")
N, T, d = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (1024, 64, 7)
X, s = synthetic_inputs(N, T, d)
X, s = X.to(dev), s.to(dev)
out = (torch.empty(N, N, device=dev), torch.empty(N, T, d, device=dev, dtype=torch.float64))


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / n * 1e3


for G in (1, 2, 4, 8):
    for fold in (True, False):
        part = [t(lambda r=r: ops.gram_sym_partial(X, 1.0, r, G, out=out, fold=fold), 10) for r in range(G)]
        Kp, gp = ops.gram_sym_partial(X, 1.0, 0, G, out=out, fold=fold)
        phi = t(lambda: ops.svgd_phi(Kp, s, gp.to(s.dtype)))

        def step():
            Kp, gp = ops.gram_sym_partial(X, 1.0, G - 1, G, out=out, fold=fold)
            return ops.svgd_phi(Kp, s, gp.to(s.dtype))

        mean = sum(part) / G
        print(f"G={G} {'folded' if fold else 'cyclic'}: partial solve per rank min {min(part):.3f} mean {mean:.3f} max "
              f"{max(part):.3f} ms (worst/mean {max(part) / mean:.3f}) | velocity {phi:.3f} ms | partial + velocity "
              f"(last rank, incl. K_partial zeroing) {t(step):.3f} ms", flush=True)
