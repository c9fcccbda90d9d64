"""Dev (GPU): a few launches of the band shapes in both modes, for rocprofv3 --kernel-trace --stats.
usage: python scripts/dev/band_modes.py N T d n [N T d n ...]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sigsvgd_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
a = [int(v) for v in sys.argv[1:]]
for k in range(0, len(a), 4):
    N, T, d, n = a[k:k + 4]
    X = torch.as_tensor(np.cumsum(0.1 * rng.standard_normal((N, T, d)), axis=1).astype(np.float32), device=dev)
    for mode in ("1", "0"):
        os.environ["SIGSVGD_BAND_MODE"] = "serial" if mode == "1" else "parallel"
        for _ in range(2):
            ops.gram_fwd_bwd(X, X, 1.0, n, y_is_x=True)
        torch.cuda.synchronize()
