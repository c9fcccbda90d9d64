import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import c_oracle as C
from oracle import sigkernel_oracle as O
from sigsvgd_amd import ops
np.set_printoptions(precision=4, linewidth=220, suppress=True)
def paths(A, T, d, seed, scale=0.05):
    rng = np.random.default_rng(seed)
    return np.cumsum(scale * rng.standard_normal((A, T, d)), axis=1).astype(np.float32)
Xp = torch.as_tensor(paths(1, 64, 7, 21)).cuda()
ops.gram_fwd_bwd(Xp, Xp, 1 / 0.8, y_is_x=True); torch.cuda.synchronize()
X, Y = paths(1, 17, 2, 21), paths(9, 17, 2, 22)
j = 0
Kr, gr = C.gram_fwd_bwd(X, Y[j:j+1], 0.8, 0)
K, g = ops.gram_fwd_bwd(torch.as_tensor(X).cuda(), torch.as_tensor(Y[j:j+1]).cuda(), 1 / 0.8)
g = g.cpu().numpy()
print("ref ch0", gr[0, :, 0]); print("gpu ch0", g[0, :, 0]); print("dif ch0", g[0, :, 0] - gr[0, :, 0])
print("ref ch1", gr[0, :, 1]); print("gpu ch1", g[0, :, 1]); print("dif ch1", g[0, :, 1] - gr[0, :, 1])
# which R[m][n] G terms would explain it: per-row contributions of each column n
Kf, gg, G = O.gram_forward_full(X.astype(np.float64), Y[j:j+1].astype(np.float64), O.RBF, 0.8, 0)
GG = O.gg_matrix(Kf, gg)
T = 17
S = GG[0, 0]
R = np.zeros((T, T)); R[1:, 1:] += S; R[:-1, :-1] += S; R[1:, :-1] -= S; R[:-1, 1:] -= S
V = O.static_grad_x(X.astype(np.float64), Y[j:j+1].astype(np.float64), G, O.RBF, 0.8)[0, 0]  # [m, n, c]
contrib = R[:, :, None] * V  # [m, n, c]
print("per-column contributions to row 0, ch0:", contrib[0, :, 0])
print("per-column contributions to row 16, ch1:", contrib[16, :, 1])
