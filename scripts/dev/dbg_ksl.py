import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sigsvgd_amd import ops
def paths(A, T, d, seed, scale=0.05):
    rng = np.random.default_rng(seed)
    return np.cumsum(scale * rng.standard_normal((A, T, d)), axis=1).astype(np.float32)
for (T, d) in [(64, 7), (40, 2), (40, 2), (17, 2), (40, 3), (40, 2)]:
    const = np.tile(np.random.default_rng(1).standard_normal((3, 1, d)).astype(np.float32), (1, T, 1))
    mov = paths(4, T, d, 2, 0.05)
    print("launch", T, d, flush=True)
    K2, g2 = ops.gram_fwd_bwd(torch.as_tensor(mov).cuda(), torch.as_tensor(const).cuda(), 1.0)
    torch.cuda.synchronize()
    print("   max |g| with a constant Y:", float(g2.abs().max()), flush=True)
