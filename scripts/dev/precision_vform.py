"""Precision experiment (numpy emulation, CPU): can the two PDE sweeps run in fp32?

Schemes, all on increments D rounded to fp32 (as gram_fast_kernel stores them) and fp32 stencil coefficients:
  f64   : the shipped scheme -- fp64 recurrence  K11 = (t - K00) + t*a + K00*b,  K_fwd / S stored fp32
  f32d  : the same delta form evaluated in fp32
  f32v  : fp32 "difference form": V[p,q] = K[p+1,q] - K[p,q] carried along the row (lane-local),
              F = t*a + K00*b ;  V += F ;  K11 = K01 + V
          rounding errors of the row recurrence are relative to |V| << |K|; the one full-magnitude add per
          cell does not feed back into V
Errors are reported for K (relative to max |K|) and for the gradient assembled in fp64 from each scheme's
S = K_fwd * U (relative to max |grad|), against the all-fp64 oracle.
usage: python scripts/dev/precision_vform.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import sigkernel_oracle as O  # noqa: E402  (dev experiment: oracle as the checker)

f32 = np.float32


def fma32(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)


def sweep(g32, scheme):
    """g32 [..., P, P] fp32 increments -> K [..., P+1, P+1] in the scheme's working precision"""
    P = g32.shape[-1]
    c12 = f32(1.0 / 12.0)
    b = (g32 * (g32 * c12)).astype(f32)
    a = fma32(g32, np.full_like(g32, 0.5), b)
    if scheme == "f64":
        K = np.ones(g32.shape[:-2] + (P + 1, P + 1))
        a64, b64 = a.astype(np.float64), b.astype(np.float64)
        for s in range(2 * P - 1):
            p = np.arange(max(0, s - P + 1), min(P, s + 1))
            q = s - p
            t = K[..., p + 1, q] + K[..., p, q + 1]
            k00 = K[..., p, q]
            K[..., p + 1, q + 1] = (t - k00) + t * a64[..., p, q] + k00 * b64[..., p, q]
        return K
    K = np.ones(g32.shape[:-2] + (P + 1, P + 1), dtype=f32)
    V = np.zeros(g32.shape[:-2] + (P,), dtype=f32)  # per row p: K[p+1,q] - K[p,q] at the row's current column
    for s in range(2 * P - 1):
        p = np.arange(max(0, s - P + 1), min(P, s + 1))
        q = s - p
        k10, k01, k00 = K[..., p + 1, q], K[..., p, q + 1], K[..., p, q]
        t = (k10 + k01).astype(f32)
        if scheme == "f32d":
            u = (t - k00).astype(f32)
            u = fma32(t, a[..., p, q], u)
            K[..., p + 1, q + 1] = fma32(k00, b[..., p, q], u)
        else:
            F = fma32(t, a[..., p, q], (k00 * b[..., p, q]).astype(f32))
            V[..., p] = (V[..., p] + F).astype(f32)
            K[..., p + 1, q + 1] = (k01 + V[..., p]).astype(f32)
    return K


def run(N, T, d, scale, h, seed=0, label=""):
    rng = np.random.default_rng(seed)
    X = np.cumsum(scale * rng.standard_normal((N, T, d)), axis=1).astype(f32)
    Kref, gref = O.gram_backward(X, X, None, O.RBF, h, 0)
    G = O.static_gram(X, X, O.RBF, h)
    D = O.increments(G)
    g32 = D.astype(f32)
    out = []
    for scheme in ("f64", "f32d", "f32v"):
        Kf = sweep(g32, scheme)
        Ur = sweep(g32[..., ::-1, ::-1], scheme)[..., ::-1, ::-1]
        Kfwd32 = Kf[..., :-1, :-1].astype(f32)
        S = (Kfwd32 * Ur[..., 1:, 1:].astype(f32)).astype(f32).astype(np.float64)
        A = X.shape[0]
        R = np.zeros((A, A, T, T))
        R[:, :, 1:, 1:] += S
        R[:, :, :-1, :-1] += S
        R[:, :, 1:, :-1] -= S
        R[:, :, :-1, 1:] -= S
        Vd = O.static_grad_x(X, X, G, O.RBF, h)
        grad = np.einsum("ijmn,ijmnc->imc", R, Vd)
        Kend = Kf[..., -1, -1].astype(f32).astype(np.float64)
        eK = np.abs(Kend - Kref).max() / np.abs(Kref).max()
        eKrel = (np.abs(Kend - Kref) / np.abs(Kref)).max()
        eg = np.abs(grad - gref).max() / np.abs(gref).max()
        out.append((scheme, eK, eKrel, eg))
    gmax = np.abs(g32).max()
    print(f"{label:34s} N={N} T={T} d={d} scale={scale} h={h}  max|g|={gmax:.3f}  K in [{Kref.min():.3g}, {Kref.max():.3g}]")
    for scheme, eK, eKrel, eg in out:
        print(f"    {scheme:5s}  K err/max|K| {eK:.2e}   max per-entry rel {eKrel:.2e}   grad err/max|grad| {eg:.2e}")


if __name__ == "__main__":
    run(12, 64, 7, 0.05, 1.0, label="C4 path shape (bench inputs)")
    run(12, 64, 3, 0.05, 1.0, label="C3 path shape")
    run(12, 32, 7, 0.05, 1.0, label="C2 path shape")
    run(8, 128, 14, 0.05, 1.0, label="C5 path shape")
    run(10, 64, 7, 0.15, 1.0, label="rougher paths")
    run(10, 64, 7, 0.3, 1.0, label="rough paths")
    run(10, 64, 7, 0.05, 0.03, label="narrow bandwidth (script h=0.03)")
    run(8, 64, 2, 0.3, 5.0, seed=3, label="wide bandwidth")
