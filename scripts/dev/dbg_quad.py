"""Diagnostics for gram_quad.hip: K / gradient error against the C oracle per case, and where along the path
(point row) the gradient error sits.  usage (GPU box): python scripts/dev/dbg_quad.py"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from oracle import c_oracle as C
from sigsvgd_amd import ops

dev = torch.device("cuda:0")
cases = [(2, 3, 128, 14, False, 0.05), (3, 2, 65, 3, False, 0.05), (3, 5, 97, 5, False, 0.3), (5, 5, 128, 14, True, 0.05),
         (4, 4, 100, 7, True, 0.05), (6, 6, 66, 2, True, 0.05), (4, 4, 128, 14, True, 0.15), (9, 9, 127, 8, True, 0.05)]
for (A, B, T, d, sym, scale) in cases:
    rng = np.random.default_rng(7)
    X = np.cumsum(scale * rng.standard_normal((A, T, d)), axis=1).astype(np.float32)
    Y = X if sym else np.cumsum(scale * rng.standard_normal((B, T, d)), axis=1).astype(np.float32)
    Kref, gref = C.gram_fwd_bwd(X, Y, 1.0, 0)
    Xg, Yg = torch.as_tensor(X, device=dev), torch.as_tensor(Y, device=dev)
    Kf = ops.gram_fwd(Xg, Xg if sym else Yg, 1.0, y_is_x=sym, stored_forward=True) if "stored_forward" in ops.gram_fwd.__code__.co_varnames else None
    K, g = ops.gram_fwd_bwd(Xg, Xg if sym else Yg, 1.0, y_is_x=sym, stored_forward=True)
    torch.cuda.synchronize()
    Kn, gn = K.cpu().numpy().astype(np.float64), g.cpu().numpy().astype(np.float64)
    ek = np.abs(Kn - Kref).max() / np.abs(Kref).max()
    eg = np.abs(gn - gref).max() / np.abs(gref).max()
    rows = np.abs(gn - gref).max(axis=(0, 2)) / np.abs(gref).max()
    bad = [(int(r), float(f"{rows[r]:.1e}")) for r in np.argsort(-rows)[:6]]
    print(f"A={A} B={B} T={T} d={d} sym={sym} scale={scale}: K {ek:.2e}  grad {eg:.2e}  finite {np.isfinite(gn).all()}  worst rows {bad}", flush=True)
