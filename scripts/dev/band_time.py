"""Dev (GPU): time the band-parallel kernel on a few shapes (timing only; used with experiment builds through SIGSVGD_LIB_PATH)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sigsvgd_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
out = []
for (N, T, d, n) in [(100, 10, 2, 4), (35, 30, 2, 3), (50, 10, 2, 4)]:
    X = torch.as_tensor(np.cumsum(0.1 * rng.standard_normal((N, T, d)), axis=1).astype(np.float32), device=dev)
    for _ in range(5):
        ops.gram_fwd_bwd(X, X, 1.0, n, y_is_x=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        ops.gram_fwd_bwd(X, X, 1.0, n, y_is_x=True)
    torch.cuda.synchronize()
    out.append(f"N={N},T={T}: {(time.perf_counter() - t0) / 50 * 1e3:.4f}")
print(os.environ.get("SIGSVGD_LIB_PATH", "in-tree").split("/")[-1], " | ".join(out), flush=True)
