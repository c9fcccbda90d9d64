import sys, numpy as np, torch
sys.path.insert(0, ".")
from oracle import c_oracle as C
from sigsvgd_amd import ops
dev = torch.device("cuda:0")
rng = np.random.default_rng(3)
for (A, T, d, n) in [(20, 64, 1, 0), (20, 33, 1, 0), (12, 128, 1, 0), (12, 100, 2, 0), (20, 33, 1, 1), (20, 17, 1, 2), (20, 64, 2, 0), (20, 5, 1, 4), (16, 9, 1, 3), (20, 20, 1, 0)]:
    for (scale, h) in [(0.02, 10.0), (0.01, 10.0), (0.02, 4.0)]:
        X = np.cumsum(scale * rng.standard_normal((A, T, d)), axis=1).astype(np.float32)
        Kref, gref = C.gram_fwd_bwd(X, X, h, n)
        Xg = torch.as_tensor(X, device=dev)
        out = []
        for sym in (True, False):
            K, g = ops.gram_fwd_bwd(Xg, Xg if sym else Xg.clone(), 1.0 / h, n, y_is_x=sym)
            eK = float((np.abs(K.double().cpu().numpy() - Kref) / np.maximum(np.abs(Kref), 1e-6)).max())
            eg = float(np.abs(g.double().cpu().numpy() - gref).max() / np.abs(gref).max())
            out.append(f"K {eK:.1e} g {eg:.1e}")
        print(f"A={A} T={T} d={d} n={n} scale={scale} h={h}: sym {out[0]} | ordered {out[1]}" + ("   <-- beyond 1e-5" if "e-05" in " ".join(out) and any(float(o.split()[3]) > 1e-5 for o in out) else ""), flush=True)
