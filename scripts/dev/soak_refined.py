"""Differential soak of the refined-grid launches (65 .. 256 cells per side) on the GPU box: random shapes, regimes and launch
forms on BOTH schedules of the band kernel (and gram_dyad.hip where SIGSVGD_BAND_MODE=serial sends a two-band grid) against the
C oracle.  usage: python scripts/dev/soak_refined.py [cases] [seed]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from oracle import c_oracle as C
from sigsvgd_amd import ops

TOL = 1e-5
shapes = [(T, n) for n in range(2, 8) for T in range(3, 34) if 64 < ((T - 1) << n) <= 256]


def main():
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    rng = np.random.default_rng(seed)
    dev = torch.device("cuda:0")
    bad = 0
    worstK = worstg = 0.0
    t0 = time.time()
    for k in range(ncases):
        T, n = shapes[int(rng.integers(0, len(shapes)))]
        d = int(rng.integers(1, 17))
        A = int(rng.integers(1, 40 if rng.random() < 0.2 else 12))
        sym = rng.random() < 0.5
        B = A if sym else int(rng.integers(1, 12))
        scale, h = [(0.05, 1.0), (0.1, 0.3), (0.3, 1.0), (0.2, 4.0), (0.5, 3.0), (0.02, 10.0)][int(rng.integers(0, 6))]
        off = 100.0 if rng.random() < 0.15 else 0.0
        X = (np.cumsum(scale * rng.standard_normal((A, T, d)), axis=1) + off).astype(np.float32)
        Y = X if sym else (np.cumsum(scale * rng.standard_normal((B, T, d)), axis=1) + off).astype(np.float32)
        use_go = rng.random() < 0.5
        go = rng.uniform(0.5, 1.5, (A, B)).astype(np.float32) if use_go else None
        Kref, gref = C.gram_fwd_bwd(X, Y, h, n, grad_out=None if go is None else go.astype(np.float64))
        if not np.isfinite(Kref).all() or np.abs(Kref).max() > 1e30:
            continue
        Xg = torch.as_tensor(X, device=dev)
        Yg = Xg if sym else torch.as_tensor(Y, device=dev)
        gog = None if go is None else torch.as_tensor(go, device=dev)
        for mode in ("serial", "parallel", ""):
            if mode:
                os.environ["SIGSVGD_BAND_MODE"] = mode
            else:
                os.environ.pop("SIGSVGD_BAND_MODE", None)
            K, g = ops.gram_fwd_bwd(Xg, Yg, 1.0 / h, n, grad_out=gog, y_is_x=sym)
            Kf = ops.gram_fwd(Xg, Yg, 1.0 / h, n, y_is_x=sym)
            eK = max(float((np.abs(t.double().cpu().numpy() - Kref) / np.maximum(np.abs(Kref), 1e-6)).max()) for t in (K, Kf))
            eg = float(np.abs(g.double().cpu().numpy() - gref).max() / max(np.abs(gref).max(), 1e-300))
            worstK, worstg = max(worstK, eK), max(worstg, eg)
            if eK < TOL and eg < TOL and max(eK, eg) > 0.6 * TOL and mode == "":
                print(f"near case {k} A={A} B={B} T={T} n={n} d={d} sym={sym} go={use_go} scale={scale} h={h} off={off}: K {eK:.2e} g {eg:.2e}", flush=True)
            if not (eK < TOL and eg < TOL):
                bad += 1
                print(f"FAIL case {k} mode={mode or 'default'} A={A} B={B} T={T} n={n} d={d} sym={sym} go={use_go} scale={scale} h={h} off={off}: K {eK:.2e} g {eg:.2e}", flush=True)
        if (k + 1) % 100 == 0:
            print(f"{k + 1} cases, {bad} failures, {time.time() - t0:.0f} s", flush=True)
    print(f"done: {ncases} cases (seed {seed}), {bad} failures; worst K entry {worstK:.2e}, worst gradient {worstg:.2e}")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
