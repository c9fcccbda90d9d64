"""EXEC discipline check: build the library with -DSIGSVGD_CHECK_EXEC (every hand-written sweep
statement group first compares EXEC with all ones and sets a sticky device word otherwise), run every kernel family that uses
the statements -- register-resident (64- and 32-slot rings, 4 / 8 / 16 channels, gradient / forward-only, symmetric / ordered,
partial shares), quadrant (both channel layouts, ROWG, EARLY, few-channel forward-only), refined-grid (4- and 8-wavefront
workgroups) -- and count the reports.
usage (GPU box):  python scripts/dev/check_exec.py build   (here: cross-compiles into sigsvgd_amd/_exp/)
                  python scripts/dev/check_exec.py run     (on the GPU box; prints the count)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "sigsvgd_amd", "_exp", "libsigsvgd_checkexec.so")

if sys.argv[1] == "build":
    from sigsvgd_amd import _lib

    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    _lib.build(out_path=LIB, objdir=os.path.join(os.path.dirname(LIB), "_obj_checkexec"), defines=("SIGSVGD_CHECK_EXEC",))
    print(LIB)
elif sys.argv[1] == "run":
    os.environ["SIGSVGD_LIB_PATH"] = LIB
    import ctypes

    import pytest

    files = ["tests/test_gpu_fast.py", "tests/test_gpu_longpaths.py", "tests/test_gpu_dyadic.py", "tests/test_gpu_precision.py",
             "tests/test_gpu_partition.py"]
    rc = pytest.main(["-q", "-m", "gpu", "-x"] + [os.path.join(ROOT, f) for f in files])
    L = ctypes.CDLL(LIB)
    words = {}
    for unit in ("fast", "quad", "dyad"):
        fn = getattr(L, f"sigsvgd_debug_exec_violations_{unit}")
        fn.restype = ctypes.c_uint
        words[unit] = fn()
    print(f"library: {LIB}\npytest exit code: {int(rc)}\nEXEC != all ones at a sweep statement (sticky word per translation unit): {words}")
    sys.exit(1 if any(words.values()) or rc else 0)
