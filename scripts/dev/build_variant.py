"""Build a variant of the library for same-box A/B runs (scripts/ab.py): python scripts/dev/build_variant.py OUT.so [-DFLAG ...]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sigsvgd_amd import _lib
out = os.path.abspath(sys.argv[1])
cmd = [_lib._hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", out] + sys.argv[2:]
cmd += [os.path.join(_lib._CSRC, s) for s in _lib.SOURCES] + ["-ldl"]
subprocess.run(cmd, check=True)
print(out)
