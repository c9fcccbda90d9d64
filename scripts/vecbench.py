"""Time the vector-kernel path (sqdist + kernel/gradient) against the reference's torch formulation
on the same GPU.  usage: python scripts/vecbench.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from sigsvgd_amd import _lib, ops

dev = torch.device("cuda:0")


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    return ms[len(ms) // 2]


def ref_gaussian(X, Y, h):
    # reference src/kernels/_kernels.py:98-111 + src/utils/math.py:69-86, verbatim formulation
    n1 = X.pow(2).sum(-1, keepdim=True)
    n2 = Y.pow(2).sum(-1, keepdim=True)
    sq = torch.addmm(n2.transpose(-2, -1), X, Y.transpose(-2, -1), alpha=-2).add_(n1).clamp(min=0)
    K = (-0.5 / h**2 * sq).exp()
    dK = -(X.unsqueeze(1) - Y) / (h**2) * K.unsqueeze(-1)
    return K, dK.sum(1)


for N, D in [(1024, 448), (4096, 64), (4096, 448), (512, 14)]:
    g = torch.Generator().manual_seed(0)
    X = torch.randn(N, D, generator=g).to(dev)
    h = float(D) ** 0.5

    def ours():
        sq = ops.vec_sqdist(X, X)
        return ops.vec_kernel(sq, X, X, _lib.VEC_GAUSSIAN, 1 / h**2, -1 / h**2)

    t_sq = timeit(lambda: ops.vec_sqdist(X, X))
    t_all = timeit(ours)
    try:
        t_ref = timeit(lambda: ref_gaussian(X, X, h), n=5) if N * N * D * 4 < 40e9 else float("nan")
    except RuntimeError:
        t_ref = float("nan")
    K, dK = ours()
    Kr, dKr = ref_gaussian(X.double(), X.double(), h) if N * N * D * 8 < 40e9 else (None, None)
    err = float((dK.double() - dKr).abs().max() / dKr.abs().max()) if dKr is not None else float("nan")
    t_fused = timeit(lambda: ops.vec_kernel_fused(X, X, _lib.VEC_GAUSSIAN, 1 / h**2, -1 / h**2))
    t_fused_atomic = timeit(lambda: ops.vec_kernel_fused(X, X, _lib.VEC_GAUSSIAN, 1 / h**2, -1 / h**2, reproducible=False))
    print(f"N={N} D={D}: fused, reproducible route (per-split partials + join) {t_fused*1e3:.1f} us; one launch with atomics {t_fused_atomic*1e3:.1f} us")
    Kf, dKf = ops.vec_kernel_fused(X, X, _lib.VEC_GAUSSIAN, 1 / h**2, -1 / h**2)
    errf = float((dKf.double() - dKr).abs().max() / dKr.abs().max()) if dKr is not None else float((dKf - dK).abs().max() / dK.abs().max())
    byf = 4 * (2 * N * D + N * N + N * D)      # fused: read X twice, write K, write dK
    fl = 2.0 * 2 * N * N * D                   # two N x N x D products
    print(f"N={N} D={D}: FUSED (one launch, fp32 MFMA) {t_fused*1e3:.1f} us = {byf/t_fused/1e6:.0f} GB/s algorithmic "
          f"({byf/t_fused/1e6/8000*100:.1f} % of 8 TB/s), {fl/t_fused/1e9:.1f} TFLOP/s ({fl/t_fused/1e9/157.3*100:.1f} % of the "
          f"157.3 TFLOP/s fp32 matrix peak), dK rel err {errf:.1e}")
    by = 4 * (2 * N * D + 3 * N * N + N * D)  # read X twice, write+read sq, write K, write dK
    print(f"N={N} D={D}: sqdist {t_sq*1e3:.1f} us, sqdist+kernel {t_all*1e3:.1f} us "
          f"({by/t_all/1e6:.1f} GB/s algorithmic), torch reference formulation {t_ref*1e3:.1f} us, dK rel err {err:.1e}")
