"""Dev check (GPU): the band-parallel kernel against the serial band kernel (SIGSVGD_BAND_SERIAL=1), bit for bit, and both
against the C oracle, over the band shapes (129 .. 256 refined cells per side)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import c_oracle as C  # noqa: E402
from sigsvgd_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
shapes = [(100, 10, 2, 4), (35, 30, 2, 3), (7, 5, 3, 6), (9, 6, 7, 5), (12, 18, 9, 3), (5, 33, 16, 3), (20, 9, 1, 5), (11, 18, 4, 3),
          (3, 3, 2, 7), (17, 25, 5, 3), (40, 13, 3, 4), (6, 31, 12, 3), (8, 21, 2, 3), (1, 10, 2, 4), (2, 12, 3, 4)]
bad = 0
if "--time-only" in sys.argv:
    shapes = []
for (N, T, d, n) in shapes:
    P = (T - 1) << n
    assert 128 < P <= 256, (T, n, P)
    for sym in (True, False):
        for grad in (True, False):
            X = np.cumsum(0.1 * rng.standard_normal((N, T, d)), axis=1).astype(np.float32)
            B = N if sym else max(1, N - 3)
            Y = X if sym else np.cumsum(0.1 * rng.standard_normal((B, T, d)), axis=1).astype(np.float32)
            go = rng.uniform(0.5, 1.5, (N, B)).astype(np.float32)
            Xg, gog = torch.as_tensor(X, device=dev), torch.as_tensor(go, device=dev)
            Yg = Xg if sym else torch.as_tensor(Y, device=dev)
            res = {}
            for mode in ("1", "0"):
                os.environ["SIGSVGD_BAND_MODE"] = "serial" if mode == "1" else "parallel"
                if grad:
                    K, g = ops.gram_fwd_bwd(Xg, Yg, 1.0, n, grad_out=gog, y_is_x=sym)
                else:
                    K, g = ops.gram_fwd(Xg, Yg, 1.0, n, y_is_x=sym), None
                torch.cuda.synchronize()
                res[mode] = (K.cpu().numpy(), None if g is None else g.cpu().numpy())
            sameK = np.array_equal(res["0"][0], res["1"][0])
            sameg = True if not grad else np.array_equal(res["0"][1], res["1"][1])
            msg = f"N={N} T={T} d={d} n={n} P={P} sym={sym} grad={grad}: K bits {'same' if sameK else 'DIFFER'} grad bits {'same' if sameg else 'DIFFER'}"
            if N * B <= 1500:
                Kref, gref = C.gram_fwd_bwd(X, Y, 1.0, n, grad_out=go.astype(np.float64))
                eK = float((np.abs(res["0"][0] - Kref) / np.maximum(np.abs(Kref), 1e-6)).max())
                msg += f"  relK {eK:.1e}"
                if grad:
                    eg = float(np.abs(res["0"][1] - gref).max() / np.abs(gref).max())
                    msg += f" relg {eg:.1e}"
                    if eg > 1e-5:
                        bad += 1
                if eK > 1e-5:
                    bad += 1
            if not (sameK and sameg):
                bad += 1
                if not sameK:
                    dK = np.argwhere(res["0"][0] != res["1"][0])
                    msg += f"  first K diffs {dK[:4].tolist()}"
            print(msg, flush=True)
# timing
for (N, T, d, n) in [(100, 10, 2, 4), (35, 30, 2, 3), (50, 10, 2, 4), (70, 10, 2, 4), (150, 10, 2, 4), (60, 30, 2, 3), (100, 30, 2, 3)]:
    X = torch.as_tensor(np.cumsum(0.1 * rng.standard_normal((N, T, d)), axis=1).astype(np.float32), device=dev)
    for mode in ("1", "0", "1", "0"):
        os.environ["SIGSVGD_BAND_MODE"] = "serial" if mode == "1" else "parallel"
        for _ in range(5):
            ops.gram_fwd_bwd(X, X, 1.0, n, y_is_x=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            ops.gram_fwd_bwd(X, X, 1.0, n, y_is_x=True)
        torch.cuda.synchronize()
        print(f"N={N} T={T} n={n} {'serial' if mode == '1' else 'parallel'}: {(time.perf_counter() - t0) / 50 * 1e3:.4f} ms per Gram + gradient", flush=True)
print("BAD", bad)
sys.exit(1 if bad else 0)
