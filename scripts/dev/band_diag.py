import sys, numpy as np, torch
sys.path.insert(0, ".")
from oracle import c_oracle as C
from sigsvgd_amd import ops
dev = torch.device("cuda:0")
for (A, B, T, n, d) in [(3, 2, 10, 4, 2), (3, 2, 5, 6, 2), (11, 9, 10, 4, 2), (3, 3, 30, 3, 2)]:
    rng = np.random.default_rng(1)
    X = np.cumsum(0.3 * rng.standard_normal((A, T, d)), axis=1).astype(np.float32)
    Y = np.cumsum(0.3 * rng.standard_normal((B, T, d)), axis=1).astype(np.float32)
    Kref, gref = C.gram_fwd_bwd(X, Y, 1.7, n)
    Xg, Yg = torch.as_tensor(X, device=dev), torch.as_tensor(Y, device=dev)
    K1 = ops.gram_fwd(Xg, Yg, 1 / 1.7, n).double().cpu().numpy()
    K2, g2 = ops.gram_fwd_bwd(Xg, Yg, 1 / 1.7, n)
    K2 = K2.double().cpu().numpy(); g2 = g2.double().cpu().numpy()
    print(f"A={A} B={B} T={T} n={n}: P={(T-1)<<n}  fwd err {np.abs(K1-Kref).max()/np.abs(Kref).max():.2e}  fwdbwd K err {np.abs(K2-Kref).max()/np.abs(Kref).max():.2e}  grad err {np.abs(g2-gref).max()/np.abs(gref).max():.2e}")
    print("   K1", K1.ravel()[:4], "ref", Kref.ravel()[:4])
