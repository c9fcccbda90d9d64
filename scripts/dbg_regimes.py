import sys, numpy as np, torch
sys.path.insert(0, '.')
from oracle import c_oracle as C
from sigsvgd_amd import ops
dev = torch.device('cuda:0')
def paths(A, T, d, seed, scale):
    rng = np.random.default_rng(seed)
    return np.cumsum(scale * rng.standard_normal((A, T, d)), axis=1).astype(np.float32)
def rel(a, b): return float(np.abs(np.asarray(a, np.float64) - b).max() / np.abs(b).max())
for (T, d) in [(128, 14), (64, 7)]:
  for (scale, h) in [(0.05, 0.1), (0.15, 1.0), (0.3, 4.0), (0.02, 0.02), (0.08, 1.0)]:
    X = paths(10, T, d, 21, scale)
    Kref, gref = C.gram_fwd_bwd(X, X, h, 0)
    Xg = torch.as_tensor(X, device=dev)
    K, g = ops.gram_fwd_bwd(Xg, Xg, 1 / h, y_is_x=True)
    Ko, g_o = ops.gram_fwd_bwd(Xg, Xg.clone(), 1 / h)
    e = np.abs(g.cpu().numpy() - gref); eo = np.abs(g_o.cpu().numpy() - gref)
    # per-particle relative error (relative to that particle's own max)
    pp = e.reshape(10, -1).max(1) / np.abs(gref).reshape(10, -1).max(1)
    ppo = eo.reshape(10, -1).max(1) / np.abs(gref).reshape(10, -1).max(1)
    offd = Kref[~np.eye(10, dtype=bool)]
    print(f"T={T} d={d} scale={scale} h={h}: Kdiag max {Kref.max():.3g} offdiag max {offd.max():.3g} | K sym {rel(K.cpu().numpy(),Kref):.1e} ord {rel(Ko.cpu().numpy(),Kref):.1e} | g sym {rel(g.cpu().numpy(),gref):.1e} ord {rel(g_o.cpu().numpy(),gref):.1e} | per-particle sym {pp.max():.1e} ord {ppo.max():.1e} | max g {np.abs(gref).max():.2g} argmax err {np.unravel_index(e.argmax(), e.shape)}")
