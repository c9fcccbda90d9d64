"""time the quadrant kernel's shapes with the library in SIGSVGD_LIB_PATH (same-box A/B): ms per Gram + gradient"""
import sys
import torch
sys.path.insert(0, ".")
from sigsvgd_amd import ops
from sigsvgd_amd.utils.synthetic import synthetic_inputs
dev = torch.device("cuda:0")
out = []
for (N, T, d, sym) in [(256, 128, 14, True), (256, 128, 14, False), (256, 100, 7, True), (256, 128, 3, True), (256, 128, 16, True)]:
    X, _ = synthetic_inputs(N, T, d)
    Xg = X.to(dev); Yg = Xg if sym else Xg.clone()
    for _ in range(3): ops.gram_fwd_bwd(Xg, Yg, 1.0, 0, y_is_x=sym)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(5):
        e0.record()
        for _ in range(4): ops.gram_fwd_bwd(Xg, Yg, 1.0, 0, y_is_x=sym)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 4)
    out.append(f"T{T}d{d}{'s' if sym else 'o'}={sorted(ts)[2]:.3f}")
print(" ".join(out))
