"""debug: per-row gradient errors of the register-resident kernel vs the C oracle for a few shapes"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import c_oracle as C
from sigsvgd_amd import ops
def paths(A, T, d, seed, scale=0.05):
    rng = np.random.default_rng(seed)
    return np.cumsum(scale * rng.standard_normal((A, T, d)), axis=1).astype(np.float32)
for (A, B, T, d) in [(1, 9, 17, 2), (1, 9, 17, 3), (1, 9, 20, 2), (2, 9, 17, 2), (1, 8, 17, 2), (1, 1, 17, 2), (1, 9, 33, 2), (1, 9, 17, 7)]:
    X, Y = paths(A, T, d, 21), paths(B, T, d, 22)
    Kr, gr = C.gram_fwd_bwd(X, Y, 0.8, 0)
    K, g = ops.gram_fwd_bwd(torch.as_tensor(X).cuda(), torch.as_tensor(Y).cuda(), 1 / 0.8)
    e = np.abs(g.cpu().numpy() - gr).max(axis=2) / np.abs(gr).max()
    print((A, B, T, d), "K err %.1e" % (np.abs(K.cpu().numpy() - Kr).max() / np.abs(Kr).max()), "grad err per point:", np.array2string(e[0], precision=1, max_line_width=250))
print("---- sequence of the failing test")
for (A, B, T, d) in [(1, 1, 64, 7), (1, 9, 17, 2), (1, 9, 17, 2)]:
    X, Y = paths(A, T, d, 21), paths(B, T, d, 22)
    Kr, gr = C.gram_fwd_bwd(X, Y, 0.8, 0)
    Xg, Yg = torch.as_tensor(X).cuda(), torch.as_tensor(Y).cuda()
    K, g = ops.gram_fwd_bwd(Xg, Yg, 1 / 0.8)
    e = np.abs(g.cpu().numpy() - gr).max(axis=2) / np.abs(gr).max()
    print((A, B, T, d), "grad err per point:", np.array2string(e[0], precision=1, max_line_width=250))
    if A == B:
        K2, g2 = ops.gram_fwd_bwd(Xg, Xg, 1 / 0.8, y_is_x=True)
        Kr2, gr2 = C.gram_fwd_bwd(X, X, 0.8, 0)
        print("   sym:", np.abs(g2.cpu().numpy() - gr2).max() / np.abs(gr2).max())
