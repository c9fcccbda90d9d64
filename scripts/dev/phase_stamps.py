"""Diagnostic build of the library with in-kernel phase stamps (-DSIGSVGD_PHASE_STAMPS): where a wave of
gram_fast_kernel spends its cycles.  Builds sigsvgd_amd/_exp/libsigsvgd_stamps.so (git-ignored scratch directory that travels to the GPU
box while it exists; never the product library; delete it after the profiling pass) and
runs a few launches; the library prints the split to stderr after each launch.
usage (on the GPU box): python scripts/dev/phase_stamps.py [N T d [sym|ordered|fwd|fwdsym|dyadic<k>]]      (build only: --build)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.environ.get("SIGSVGD_STAMPS_LIB") or os.path.join(ROOT, "sigsvgd_amd", "_exp", "libsigsvgd_stamps.so")


def build():
    from sigsvgd_amd import _lib

    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    _lib.build(out_path=OUT, objdir=os.path.join(os.path.dirname(OUT), "_obj_stamps"), defines=("SIGSVGD_PHASE_STAMPS",))


if __name__ == "__main__":
    if "--build" in sys.argv:
        build()
        sys.exit(0)
    if os.environ.get("SIGSVGD_LIB_PATH") != OUT:
        if not os.path.exists(OUT):
            build()
        env = dict(os.environ, SIGSVGD_LIB_PATH=OUT)
        sys.exit(subprocess.run([sys.executable] + sys.argv, env=env).returncode)
    import torch

    from sigsvgd_amd import ops
    from sigsvgd_amd.utils.synthetic import synthetic_inputs

    n, t, d = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (1024, 64, 7)
    mode = sys.argv[4] if len(sys.argv) >= 5 else "sym"  # sym | ordered | fwd | fwdsym | dyadic<k> (coverage kernel)
    if mode.startswith("dyadic"):
        X, _ = synthetic_inputs(n, t, d)
        Xg = X.cuda()
        for _ in range(3):
            ops.gram_fwd_bwd(Xg, Xg, 1.0, int(mode[6:]), y_is_x=True)
        torch.cuda.synchronize()
        sys.exit(0)
    X, _ = synthetic_inputs(n, t, d)
    Xg = X.cuda()
    for _ in range(3):
        if t > 64:
            ops.gram_fwd_bwd(Xg, Xg, 1.0, 0, y_is_x=(mode == "sym"), stored_forward=True)
        elif mode == "sym":
            ops.gram_fwd_bwd(Xg, Xg, 1.0, 0, y_is_x=True)
        elif mode == "ordered":
            ops.gram_fwd_bwd(Xg, Xg, 1.0, 0)
        elif mode == "fwd":
            ops.gram_fwd(Xg, Xg, 1.0, 0)
        else:
            ops.gram_fwd(Xg, Xg, 1.0, 0, y_is_x=True)
    torch.cuda.synchronize()
