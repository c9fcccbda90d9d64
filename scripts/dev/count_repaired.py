"""How many entries of K does the fp64 pass change?  Same inputs through two builds (SIGSVGD_LIB_PATH), bitwise diff."""
import os, subprocess, sys, tempfile
import torch
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ".")
    from sigsvgd_amd.utils.synthetic import synthetic_inputs
    from sigsvgd_amd import ops
    N, T, d = (int(v) for v in sys.argv[3:6])
    X, _ = synthetic_inputs(N, T, d)
    K = ops.gram_fwd(X.cuda(), X.cuda(), 1.0, y_is_x=True)
    torch.save(K.cpu(), sys.argv[2])
    sys.exit(0)
libA, libB = sys.argv[1:3]
for shape in sys.argv[3:]:
    N, T, d = shape.split(",")
    outs = []
    for lib in (libA, libB):
        f = tempfile.mktemp(suffix=".pt")
        subprocess.run([sys.executable, __file__, "--child", f, N, T, d], env=dict(os.environ, SIGSVGD_LIB_PATH=os.path.abspath(lib)), check=True)
        outs.append(torch.load(f))
    diff = (outs[0] != outs[1])
    rel = ((outs[0] - outs[1]).abs() / outs[1].abs().clamp_min(0.1)).max().item()
    print(f"N={N} T={T} d={d}: {int(diff.sum())} of {diff.numel()} entries differ, largest relative difference {rel:.2e}; K range {outs[1].min().item():.3g} .. {outs[1].max().item():.3g}", flush=True)
