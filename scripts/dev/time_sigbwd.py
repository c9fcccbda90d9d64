"""signature adjoint vs forward, microseconds (GPU box)"""
import sys
import torch
sys.path.insert(0, ".")
from sigsvgd_amd import ops
g = torch.Generator().manual_seed(0)
dev = torch.device("cuda:0")
for (N, L, C, depth) in [(1024, 64, 2, 3), (1024, 64, 2, 2), (1024, 64, 3, 2), (1024, 64, 4, 2), (100, 10, 2, 3), (1024, 64, 2, 4), (1024, 64, 3, 3), (1024, 64, 5, 2), (1024, 64, 6, 2), (256, 32, 3, 4), (64, 100, 7, 3), (128, 50, 2, 6)]:
    P = torch.cumsum(0.3 * torch.randn(N, L, C, generator=g), 1).to(dev)
    S = ops.signature(P, depth, basepoint=True)
    gs = torch.randn(S.shape, generator=g).to(dev)
    for _ in range(3): ops.signature_backward(P, gs, depth, basepoint=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.signature_backward(P, gs, depth, basepoint=True)
    e1.record(); torch.cuda.synchronize()
    f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    f0.record()
    for _ in range(10): ops.signature(P, depth, basepoint=True)
    f1.record(); torch.cuda.synchronize()
    print(N, L, C, depth, "backward %.1f us, forward %.1f us" % (e0.elapsed_time(e1) * 100, f0.elapsed_time(f1) * 100))
