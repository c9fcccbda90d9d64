import sys
import numpy as np, torch
sys.path.insert(0, ".")
from oracle import sigkernel_oracle as O
from sigsvgd_amd import ops
dev = torch.device("cuda:0")
def paths(A, T, d, seed, scale=0.05):
    rng = np.random.default_rng(seed)
    return np.cumsum(scale * rng.standard_normal((A, T, d)), axis=1).astype(np.float32)
for (N, T, n, d) in [(19, 17, 2, 14), (19, 17, 2, 8), (19, 17, 2, 9), (19, 17, 2, 16), (64, 17, 2, 14), (19, 10, 4, 14), (5, 17, 2, 14)]:
    X = paths(N, T, d, 5)
    Kref, gref = O.gram_backward(X, X, None, O.RBF, 0.9, n)
    Xg = torch.as_tensor(X, device=dev)
    K, g = ops.gram_fwd_bwd(Xg, Xg, 1 / 0.9, n, y_is_x=True)
    K2, g2 = ops.gram_fwd_bwd(Xg, Xg.clone(), 1 / 0.9, n)
    e = np.abs(g.cpu().numpy() - gref) / np.abs(gref).max()
    e2 = np.abs(g2.cpu().numpy() - gref) / np.abs(gref).max()
    print(N, T, n, d, "sym err", e.max(), "ordered err", e2.max(), "K err", np.abs(K.cpu().numpy() - Kref).max())
    if e.max() > 1e-4:
        print("  per row max:", np.round(e.max(axis=(1, 2)), 3))
        print("  per t max:", np.round(e.max(axis=(0, 2)), 3))
        print("  per c max:", np.round(e.max(axis=(0, 1)), 3))
