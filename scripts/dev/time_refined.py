"""time the refined-grid / band kernels' shapes with the library in SIGSVGD_LIB_PATH: ms per Gram + gradient (symmetric), forward only"""
import sys
import torch
sys.path.insert(0, ".")
from sigsvgd_amd import ops
from sigsvgd_amd.utils.synthetic import synthetic_inputs
dev = torch.device("cuda:0")
out = []
for (N, T, d, n) in [(100, 10, 2, 4), (35, 30, 2, 3), (16, 20, 2, 2), (30, 5, 2, 5), (256, 10, 2, 4)]:
    X, _ = synthetic_inputs(N, T, d)
    Xg = X.to(dev)
    def t(fn):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for _ in range(5):
            e0.record()
            for _ in range(10): fn()
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 10)
        return sorted(ts)[2]
    out.append(f"N{N}T{T}n{n}: {t(lambda: ops.gram_fwd_bwd(Xg, Xg, 1.0, n, y_is_x=True)):.4f}/{t(lambda: ops.gram_fwd(Xg, Xg, 1.0, n, y_is_x=True)):.4f}")
print("  ".join(out))
