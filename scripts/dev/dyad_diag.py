import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import c_oracle as C
from sigsvgd_amd import ops
dev = torch.device('cuda:0')
def paths(A, T, d, seed, scale=0.3):
    rng = np.random.default_rng(seed)
    return np.cumsum(scale * rng.standard_normal((A, T, d)), axis=1).astype(np.float32)
C.build()
for (T, n, d, h, scale) in [(30, 2, 4, 0.9, 0.3), (30, 2, 4, 0.9, 0.1), (20, 2, 2, 1.0, 0.3), (5, 5, 2, 0.9, 0.3), (5, 5, 2, 0.3, 0.5), (3, 6, 7, 0.9, 0.3), (33, 2, 5, 0.9, 0.3), (33, 2, 5, 0.3, 0.3), (17, 2, 14, 0.9, 0.3)]:
    X = paths(19, T, d, 5, scale)
    Kref, gref = C.gram_fwd_bwd(X, X, h, n)
    Xg = torch.as_tensor(X, device=dev)
    out = []
    for name, kw in [('dyad sym', dict(y_is_x=True)), ('dyad ord', dict()), ('generic', dict(force_generic=True))]:
        K, g = ops.gram_fwd_bwd(Xg, Xg.clone() if 'ord' in name else Xg, 1.0 / h, n, **kw)
        Kn = K.double().cpu().numpy()
        e = np.abs(Kn - Kref) / np.abs(Kref)
        ij = np.unravel_index(e.argmax(), e.shape)
        ge = np.abs(g.double().cpu().numpy() - gref).max() / np.abs(gref).max()
        out.append(f"{name}: K {e.max():.1e} at {ij} (K={Kref[ij]:.3g}) grad {ge:.1e}")
    print(f"T={T} n={n} d={d} h={h} scale={scale} Kmax={Kref.max():.3g}: " + " | ".join(out), flush=True)
