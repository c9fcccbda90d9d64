"""A/B timing of experimental builds of the library (compile-time switches such as -DSIGQ_NW=4 in gram_quad.hip;
timing only).  Build here:  python scripts/dev/ab_quad.py --build NAME=-DFLAG[,-DFLAG2] ...   (NAME= alone: no flag)
run on the GPU box:  python scripts/dev/ab_quad.py NAME ... [-- N T d]      (long-path shape, default 256 128 14)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
DIR = os.path.join(ROOT, "sigsvgd_amd", "_exp")


def lib(k):
    return os.path.join(DIR, f"libsigsvgd_exp{k}.so")


def build(k, extra=()):
    from sigsvgd_amd import _lib

    os.makedirs(DIR, exist_ok=True)
    cmd = [_lib._hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", *extra, "-o", lib(k)] + [os.path.join(_lib._CSRC, s) for s in _lib.SOURCES] + ["-ldl"]
    subprocess.run(cmd, check=True)


def child(shape):
    import torch

    from sigsvgd_amd import ops
    from sigsvgd_amd.utils.synthetic import synthetic_inputs

    n, t, d = shape
    X, _ = synthetic_inputs(n, t, d)
    Xg = X.cuda()
    for sym in (True, False):
        for _ in range(3):
            ops.gram_fwd_bwd(Xg, Xg, 1.0, 0, y_is_x=sym)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.gram_fwd_bwd(Xg, Xg, 1.0, 0, y_is_x=sym)
        e1.record()
        torch.cuda.synchronize()
        print(f"  {'sym' if sym else 'ordered'} {e0.elapsed_time(e1) / 10:.3f} ms", end="")
    for sym in (True, False):
        for _ in range(3):
            ops.gram_fwd(Xg, Xg, 1.0, 0, y_is_x=sym)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.gram_fwd(Xg, Xg, 1.0, 0, y_is_x=sym)
        e1.record()
        torch.cuda.synchronize()
        print(f"  fwd {'sym' if sym else 'ordered'} {e0.elapsed_time(e1) / 10:.3f} ms", end="")
    print(flush=True)


if __name__ == "__main__":
    args = sys.argv[1:]
    shape = (256, 128, 14)
    if "--" in args:
        k = args.index("--")
        shape = tuple(int(v) for v in args[k + 1:k + 4])
        args = args[:k]
    if args and args[0] == "--build":
        for spec in args[1:]:
            name, _, flags = spec.partition("=")
            build(name, [f for f in flags.split(",") if f])
        sys.exit(0)
    if args and args[0] == "--child":
        child(shape)
        sys.exit(0)
    for k in args:
        print(f"{k} N,T,d={shape}:", end="", flush=True)
        env = dict(os.environ, SIGSVGD_LIB_PATH=lib(k))
        subprocess.run([sys.executable, __file__, "--child", "--", *map(str, shape)], env=env, check=True)
