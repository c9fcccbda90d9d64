"""CPU study (numpy emulation; dev experiment, the oracle is the checker): which per-pair quantity predicts the error of the
fp32 route (fp32-stored increments, fp32 difference-form sweeps) on rough paths in few channels?

Replays the generator of scripts/dev/soak.py (same seed, same draws) up to the requested cases, then for every pair:
  err   = |K_emul - K_ref| / max(|K_ref|, 0.1)          (emulated fp32 route vs all-fp64)
  err64 = the same for fp64 sweeps on fp32 increments   (what the in-kernel re-sweep returns)
  kmax  = max |K_grid| / max(|K|, 0.1)                  (the flag of round 3)
  c1    = sum |S*D| / max(|K|, 0.1)                     (first-order condition number of K w.r.t. relative errors of D)
  c2    = sqrt(sum (S*D)^2) / max(|K|, 0.1)
usage: python scripts/dev/cond_study.py case [case ...]   (soak case numbers, seed 7)
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import sigkernel_oracle as O  # noqa: E402
from scripts.dev.precision_vform import sweep  # noqa: E402

f32 = np.float32


def soak_cases(wanted, seed=7):
    """yield (k, X, Y, h, n) for the wanted case numbers, replaying soak.py's draws"""
    rng = np.random.default_rng(seed)
    last = max(wanted)
    for k in range(last + 1):
        T = int(rng.choice([3, 5, 8, 13, 16, 20, 31, 32, 33, 47, 64, 65, 70, 100, 128]))
        d = int(rng.integers(1, 17))
        big = rng.random() < 0.25
        A = int(rng.integers(1, 90 if big and T <= 64 else 20))
        B = A if rng.random() < 0.5 else int(rng.integers(1, 90 if big and T <= 64 else 20))
        n = int(rng.choice([0, 0, 1, 2, 3, 4])) if T <= 20 else 0
        if n >= 3 and T > 8:
            n = 2
        if rng.random() < 0.25:
            T, n = [(3, 5), (3, 6), (5, 4), (5, 5), (9, 3), (9, 4), (17, 2), (17, 3), (20, 2), (33, 1), (33, 2), (12, 3),
                    (10, 4), (30, 3), (33, 3), (3, 7), (20, 3), (5, 6), (18, 3), (17, 4)][int(rng.integers(0, 20))]
        h = float(rng.choice([0.3, 1.0, 4.0]))
        scale = 0.05 if T > 64 else 0.08
        if rng.random() < 0.25:
            d = int(rng.integers(1, 4))
            scale, h = [(0.2, 0.1), (0.5, 1.0), (0.1, 0.1), (0.3, 0.3)][int(rng.integers(0, 4))]
        X = np.cumsum(scale * rng.standard_normal((A, T, d)), axis=1).astype(np.float32)
        Y = X if A == B and rng.random() < 0.5 else np.cumsum(scale * rng.standard_normal((B, T, d)), axis=1).astype(np.float32)
        yx = Y is X
        rng.uniform(0.5, 1.5, (A, B))
        rng.random()  # (the forced-coverage-kernel draw)
        if yx and n == 0 and 3 <= T <= 128 and rng.random() < 0.5:
            rng.integers(1, 6)
            rng.random()
        if k in wanted:
            yield k, X, Y, h, n, scale


def study(X, Y, h, n=0, label=""):
    G = O.static_gram(X, Y, O.RBF, h)
    D = O.refine(O.increments(G), n)
    Kf = O.pde_sweep(D)
    Ur = O.pde_sweep(D[..., ::-1, ::-1])[..., ::-1, ::-1]
    Kref = Kf[..., -1, -1]
    S = Kf[..., :-1, :-1] * Ur[..., 1:, 1:]
    den = np.maximum(np.abs(Kref), 0.1)
    g32 = D.astype(f32)
    K32 = sweep(g32, "f32v")[..., -1, -1].astype(np.float64)
    K64 = sweep(g32, "f64")[..., -1, -1]
    err = np.abs(K32 - Kref) / den
    err64 = np.abs(K64 - Kref) / den
    kmax = np.abs(Kf).max(axis=(-1, -2)) / den
    c1 = np.abs(S * D).sum(axis=(-1, -2)) / den
    c2 = np.sqrt(((S * D) ** 2).sum(axis=(-1, -2))) / den
    sD = np.abs(D).sum(axis=(-1, -2))
    return dict(err=err, err64=err64, kmax=kmax, c1=c1, c2=c2, sD=sD, K=Kref, gmax=np.abs(D).max(axis=(-1, -2)))


def report(label, r):
    e = r["err"].ravel()
    order = np.argsort(-e)[:5]
    print(f"{label}: pairs {e.size}  max err {e.max():.2e}  (fp64 sweeps on fp32 increments: {r['err64'].max():.2e})")
    for o in order:
        print(f"    err {e[o]:.2e} err64 {r['err64'].ravel()[o]:.2e}  K {r['K'].ravel()[o]:+.3e}  kmax/K {r['kmax'].ravel()[o]:8.2f}  "
              f"c1 {r['c1'].ravel()[o]:9.1f}  c2 {r['c2'].ravel()[o]:8.2f}  sum|D| {r['sD'].ravel()[o]:8.1f} max|D| {r['gmax'].ravel()[o]:.3f}")
    # how good are the predictors: among pairs with err > 3e-6, the smallest value of each; among pairs with err < 1e-6, the largest
    hi, lo = e > 3e-6, e < 1e-6
    for name in ("kmax", "c1", "c2"):
        v = r[name].ravel()
        print(f"    {name:5s} min over err>3e-6: {v[hi].min() if hi.any() else float('nan'):10.2f}   max over err<1e-6: "
              f"{v[lo].max() if lo.any() else float('nan'):10.2f}   corr(log err, log {name}) {np.corrcoef(np.log(e + 1e-12), np.log(v + 1e-12))[0, 1]:.2f}")


if __name__ == "__main__":
    wanted = [int(a) for a in sys.argv[1:]] or [213, 492, 1118, 1995, 2383, 2441, 3027, 3526, 4339]
    for k, X, Y, h, n, scale in soak_cases(set(wanted)):
        A, T, d = X.shape
        if A * Y.shape[0] > 1200:  # keep the emulation in seconds
            X, Y = X[:30], Y[:30] if Y is not X else X[:30]
        r = study(X, Y, h, n)
        report(f"case {k} A={X.shape[0]} B={Y.shape[0]} T={T} d={d} n={n} h={h} scale={scale}", r)
    # smooth references: the bench inputs
    for (N, T, d) in [(12, 64, 3), (12, 64, 7), (8, 128, 14), (12, 32, 7), (12, 64, 1), (12, 64, 2)]:
        Xb, _ = O.synthetic_inputs(N, T, d)
        Xb = Xb.numpy()
        report(f"bench inputs N={N} T={T} d={d}", study(Xb, Xb, 1.0))
