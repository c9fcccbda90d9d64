import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import c_oracle as C
from sigsvgd_amd import ops
def paths(A, T, d, seed, scale=0.05):
    rng = np.random.default_rng(seed)
    return np.cumsum(scale * rng.standard_normal((A, T, d)), axis=1).astype(np.float32)
Xp = torch.as_tensor(paths(1, 64, 7, 21)).cuda()
X, Y = paths(1, 17, 2, 21), paths(9, 17, 2, 22)
Xg, Yg = torch.as_tensor(X).cuda(), torch.as_tensor(Y).cuda()
for rep in range(3):
    ops.gram_fwd_bwd(Xp, Xp, 1 / 0.8, y_is_x=True); torch.cuda.synchronize()
    for j in range(9):
        Kr, gr = C.gram_fwd_bwd(X, Y[j:j+1], 0.8, 0)
        K, g = ops.gram_fwd_bwd(Xg, Yg[j:j+1].contiguous(), 1 / 0.8)
        e = np.abs(g.cpu().numpy() - gr).max(axis=2)[0] / np.abs(gr).max()
        print(rep, j, "max err %.1e" % e.max(), "rows>1e-5:", np.nonzero(e > 1e-5)[0].tolist(), flush=True)
