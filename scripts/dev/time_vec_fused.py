"""fused vector kernel, microseconds per call (GPU box): N x D shapes, Gaussian, reproducible route (+ join) and the kernel alone"""
import sys
import torch
sys.path.insert(0, ".")
from sigsvgd_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for (N, D) in [(1024, 448), (4096, 64), (4096, 448), (2048, 448), (1024, 64), (100, 20)]:
    V = torch.randn(N, D, generator=g).to(dev)
    for rep in (True, False):
        for _ in range(3):
            ops.vec_kernel_fused(V, V, 0, 1.0 / (2.0 * D), 1.0, reproducible=rep)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.vec_kernel_fused(V, V, 0, 1.0 / (2.0 * D), 1.0, reproducible=rep)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 50
        print(f"N={N} D={D} reproducible={rep}: {us:.1f} us  ({4.0 * N * N * D / us / 1e6:.1f} TFLOP/s)", flush=True)
