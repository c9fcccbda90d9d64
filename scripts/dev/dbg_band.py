import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import c_oracle as C
from sigsvgd_amd import ops
np.set_printoptions(precision=2, linewidth=250)
def paths(A, T, d, seed, scale=0.05):
    rng = np.random.default_rng(seed)
    return np.cumsum(scale * rng.standard_normal((A, T, d)), axis=1).astype(np.float32)
for (A, B, T, d, sym) in [(1, 1, 128, 14, False), (2, 3, 128, 14, False), (5, 5, 128, 14, True), (3, 2, 65, 3, False), (4, 4, 100, 7, True), (6, 6, 66, 2, True)]:
    X = paths(A, T, d, 1); Y = X if sym else paths(B, T, d, 2)
    Kr, gr = C.gram_fwd_bwd(X, Y, 1.0, 0)
    Xg, Yg = torch.as_tensor(X).cuda(), torch.as_tensor(Y).cuda()
    Kf = ops.gram_fwd(Xg, Yg, 1.0, y_is_x=sym)
    K, g = ops.gram_fwd_bwd(Xg, Xg if sym else Yg, 1.0, y_is_x=sym, check_regime=False)
    torch.cuda.synchronize()
    eK = np.abs(K.cpu().numpy() - Kr).max() / np.abs(Kr).max(); eKf = np.abs(Kf.cpu().numpy() - Kr).max() / np.abs(Kr).max()
    e = np.abs(g.cpu().numpy() - gr).max(axis=2) / np.abs(gr).max()
    print((A, B, T, d, sym), "K err %.1e (fwd-only %.1e)" % (eK, eKf), "grad err max %.1e" % e.max(), "worst rows of particle 0:", np.argsort(-e[0])[:6], e[0][np.argsort(-e[0])[:6]], flush=True)
