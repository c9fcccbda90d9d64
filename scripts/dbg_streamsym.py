import sys, numpy as np, torch
sys.path.insert(0, '.')
from oracle import c_oracle as C
from sigsvgd_amd import ops
dev = torch.device('cuda:0')
def paths(A, T, d, seed, scale=0.05):
    rng = np.random.default_rng(seed)
    return np.cumsum(scale * rng.standard_normal((A, T, d)), axis=1).astype(np.float32)
def rel(a, b): return float(np.abs(np.asarray(a, np.float64) - b).max() / np.abs(b).max())
N, T, d, h = 9, 128, 14, 0.9
X = paths(N, T, d, 5)
go = np.random.default_rng(6).standard_normal((N, N)).astype(np.float32)
Xg, gog = torch.as_tensor(X, device=dev), torch.as_tensor(go, device=dev)
for name, w in [("random", go), ("ones", np.ones_like(go)), ("offdiag", go * (1 - np.eye(N, dtype=np.float32))), ("diag", go * np.eye(N, dtype=np.float32))]:
    Kref, gref = C.gram_fwd_bwd(X, X, h, 0, grad_out=w.astype(np.float64))
    wg = torch.as_tensor(w, device=dev)
    K, g = ops.gram_fwd_bwd(Xg, Xg, 1 / h, grad_out=wg, y_is_x=True)
    Ko, g_o = ops.gram_fwd_bwd(Xg, Xg, 1 / h, grad_out=wg)
    e = np.abs(g.cpu().numpy() - gref); eo = np.abs(g_o.cpu().numpy() - gref)
    print(name, "sym", rel(g.cpu().numpy(), gref), "ordered", rel(g_o.cpu().numpy(), gref), "max|g|", np.abs(gref).max(),
          "argmax err sym", np.unravel_index(e.argmax(), e.shape), "ord", np.unravel_index(eo.argmax(), eo.shape))
w = go * np.eye(N, dtype=np.float32)
Kref, gref = C.gram_fwd_bwd(X, X, h, 0, grad_out=w.astype(np.float64))
K, g = ops.gram_fwd_bwd(Xg, Xg, 1 / h, grad_out=torch.as_tensor(w, device=dev))
e = np.abs(g.cpu().numpy() - gref)[8].max(1) / np.abs(gref).max()
print("per-row err (x1e6), pair (8,8):", np.round(e[56:72] * 1e6, 2))
print("rows max:", np.round(np.sort(e)[-5:] * 1e6, 2), np.argsort(e)[-5:])
# force generic for comparison
Kg, gg = ops.gram_fwd_bwd(Xg, Xg, 1 / h, grad_out=torch.as_tensor(w, device=dev), force_generic=True)
eg = np.abs(gg.cpu().numpy() - gref)[8].max(1) / np.abs(gref).max()
print("generic rows max:", np.round(np.sort(eg)[-5:] * 1e6, 2), np.argsort(eg)[-5:])
er = (g.cpu().numpy() - gref)[8, 64]
xt = (X[8, 64] - X[8, 0]).astype(np.float64)
print("err row64:", np.round(er * 1e3, 3))
print("x~ row64 :", np.round(xt, 3))
print("ratio    :", np.round(er / xt * 1e3, 3))
er63 = (g.cpu().numpy() - gref)[8, 63]; er65 = (g.cpu().numpy() - gref)[8, 65]
print("err row63:", np.round(er63 * 1e3, 3)); print("err row65:", np.round(er65 * 1e3, 3))
