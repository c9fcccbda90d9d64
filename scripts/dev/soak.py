"""One-off differential soak on the GPU box: many random shapes (all three solvers, ordered / symmetric / forward-only /
partial shares, dyadic orders 0..4) against the C oracle.  usage: python scripts/dev/soak.py [cases] [seed]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from oracle import c_oracle as C
from sigsvgd_amd import ops

TOL = 1e-5


def rel(a, b):
    return float(np.abs(np.asarray(a, np.float64) - b).max() / max(np.abs(b).max(), 1e-300))


def main():
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
    dev = torch.device("cuda:0")
    bad = 0
    t0 = time.time()
    for k in range(ncases):
        T = int(rng.choice([3, 5, 8, 13, 16, 20, 31, 32, 33, 47, 64, 65, 70, 100, 128]))
        d = int(rng.integers(1, 17))
        big = rng.random() < 0.25
        A = int(rng.integers(1, 90 if big and T <= 64 else 20))
        B = A if rng.random() < 0.5 else int(rng.integers(1, 90 if big and T <= 64 else 20))
        n = int(rng.choice([0, 0, 1, 2, 3, 4])) if T <= 20 else 0
        if n >= 3 and T > 8:
            n = 2
        h = float(rng.choice([0.3, 1.0, 4.0]))
        scale = 0.05 if T > 64 else 0.08
        X = np.cumsum(scale * rng.standard_normal((A, T, d)), axis=1).astype(np.float32)
        Y = X if A == B and rng.random() < 0.5 else np.cumsum(scale * rng.standard_normal((B, T, d)), axis=1).astype(np.float32)
        yx = Y is X
        go = rng.uniform(0.5, 1.5, (A, B)).astype(np.float32)
        Kref, gref = C.gram_fwd_bwd(X, Y, h, n, grad_out=go.astype(np.float64))
        Xg, gog = torch.as_tensor(X, device=dev), torch.as_tensor(go, device=dev)
        Yg = Xg if yx else torch.as_tensor(Y, device=dev)
        errs = {}
        K, g = ops.gram_fwd_bwd(Xg, Yg, 1.0 / h, n, grad_out=gog, y_is_x=yx)
        errs["K"], errs["g"] = rel(K.cpu().numpy(), Kref), rel(g.cpu().numpy(), gref)
        errs["Kfwd"] = rel(ops.gram_fwd(Xg, Yg, 1.0 / h, n, y_is_x=yx).cpu().numpy(), Kref)
        if T <= 100 and rng.random() < 0.3: # (the coverage kernel's per-pair state has to fit 160 KB of LDS)
            K3, g3 = ops.gram_fwd_bwd(Xg, Yg, 1.0 / h, n, grad_out=gog, y_is_x=yx, force_generic=True)
            errs["Kgen"], errs["ggen"] = rel(K3.cpu().numpy(), Kref), rel(g3.cpu().numpy(), gref)
        if yx and n == 0 and 3 <= T <= 128 and rng.random() < 0.5:
            stride = int(rng.integers(1, 6))
            Ks = torch.zeros(A, A, device=dev)
            gs = torch.zeros(A, T, d, device=dev, dtype=torch.float64)
            for r in range(stride):
                Kp, gp = ops.gram_sym_partial(Xg, 1.0 / h, r, stride, grad_out=gog)
                Ks += Kp
                gs += gp
            errs["Kpart"], errs["gpart"] = rel(Ks.cpu().numpy(), Kref), rel(gs.cpu().numpy(), gref)
        worst = max(errs.values())
        if not np.isfinite(worst) or worst > TOL:
            bad += 1
            print(f"FAIL case {k}: A={A} B={B} T={T} d={d} n={n} h={h} yx={yx}: {errs}", flush=True)
        if k % 50 == 49:
            print(f"{k + 1} cases, {bad} failures, {time.time() - t0:.0f} s", flush=True)
    print(f"done: {ncases} cases, {bad} failures")
    sys.exit(1 if bad else 0)


main()
