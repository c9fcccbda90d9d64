"""Dev (GPU): 65 .. 128-cell shapes at growing N: the launcher's default, the refined-grid kernel (SIGSVGD_BAND_MODE=serial) and
the band kernel's band-parallel schedule (=parallel); Gram + gradient, symmetric, ms."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sigsvgd_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
for (N, T, d, n) in [(64, 20, 2, 2), (150, 20, 2, 2), (300, 20, 2, 2), (100, 9, 2, 4), (200, 5, 2, 5), (400, 5, 2, 5), (200, 17, 3, 3), (128, 33, 7, 2)]:
    X = torch.as_tensor(np.cumsum(0.1 * rng.standard_normal((N, T, d)), axis=1).astype(np.float32), device=dev)
    out = []
    for mode in ("", "serial", "parallel"):
        if mode:
            os.environ["SIGSVGD_BAND_MODE"] = mode
        else:
            os.environ.pop("SIGSVGD_BAND_MODE", None)
        for _ in range(3):
            ops.gram_fwd_bwd(X, X, 1.0, n, y_is_x=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            ops.gram_fwd_bwd(X, X, 1.0, n, y_is_x=True)
        torch.cuda.synchronize()
        out.append(f"{mode or 'default'} {(time.perf_counter() - t0) / 20 * 1e3:.4f}")
    print(f"N={N} T={T} d={d} n={n} P={(T - 1) << n} pairs={N * (N + 1) // 2}: " + " | ".join(out), flush=True)
