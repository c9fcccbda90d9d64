"""ctypes binding of libsigsvgd_hip.so (C ABI in include/sigsvgd_hip.h) + in-tree build helper.

There is deliberately NO fallback: if the shared library is missing or fails to load, every hot-path
op raises.  `build()` cross-compiles for gfx950 with hipcc (works without a GPU).
"""
from __future__ import annotations

import ctypes
import os
import shutil
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_PKG, "csrc")
# SIGSVGD_LIB_PATH: A/B benchmarking of two builds on the same GPU box (scripts/ab.py); never set in tests
LIB_PATH = os.environ.get("SIGSVGD_LIB_PATH") or os.path.join(_PKG, "libsigsvgd_hip.so")
SOURCES = ["capi.hip", "gram_generic.hip", "gram_fast.hip", "gram_quad.hip", "svgd_phi.hip",
           "vec_kernels.hip", "vec_fused.hip", "cost_kernels.hip", "sig_backward.hip", "gram_dyad.hip", "gram_band.hip"]
HEADERS = [os.path.join(_CSRC, "sig_common.h"), os.path.join(_CSRC, "quad_sweeps.h"),
           os.path.join(_PKG, "..", "include", "sigsvgd_hip.h")]

# mirror of include/sigsvgd_hip.h
F32, F64 = 0, 1
STATIC_RBF, STATIC_LINEAR = 0, 1
FLAG_NAIVE_SOLVER, FLAG_SYM, FLAG_Y_IS_X, FLAG_FORCE_GENERIC, FLAG_WS_CLEAN, FLAG_STORED_FORWARD = 1, 2, 4, 8, 16, 32
FLAG_FOLD_TILES = 64
VEC_GAUSSIAN, VEC_IMQ, VEC_UNIT = 0, 1, 2
ABI_VERSION = 9

EXPORTS = [
    "sigsvgd_abi_version",
    "sigsvgd_last_error",
    "sigsvgd_gram_workspace_bytes",
    "sigsvgd_gram_fwd",
    "sigsvgd_gram_fwd_bwd",
    "sigsvgd_gram_sym_partial",
    "sigsvgd_gram_sym_tile_rows",
    "sigsvgd_svgd_phi",
    "sigsvgd_svgd_step",
    "sigsvgd_svgd_adam_step",
    "sigsvgd_vec_sqdist",
    "sigsvgd_vec_kernel",
    "sigsvgd_vec_kernel_fused",
    "sigsvgd_vec_fused_workspace_bytes",
    "sigsvgd_signature",
    "sigsvgd_signature_backward",
    "sigsvgd_obstacle_cost",
]

_lib = None


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm's hipcc to build libsigsvgd_hip.so)")


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(_CSRC, s) for s in SOURCES] + HEADERS
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force: bool = False, verbose: bool = False, out_path: str = None, objdir: str = None, defines=()) -> str:
    """hipcc --offload-arch=gfx950 -shared -> sigsvgd_amd/libsigsvgd_hip.so (in-tree).  `out_path` / `objdir`: build somewhere
    else from scratch (tests/test_cabi.py builds into a temporary directory to show the sources alone produce the library;
    scripts/dev builds diagnostic variants with `defines`)."""
    if out_path is not None:
        return _build_to(out_path, objdir or os.path.join(os.path.dirname(out_path), "_obj"), verbose, tuple(defines))
    if not force and not needs_build():
        return LIB_PATH
    return _build_to(LIB_PATH, os.path.join(_PKG, "_obj"), verbose, tuple(defines))


def _build_to(lib_path: str, objdir: str, verbose: bool, defines=()) -> str:
    # one hipcc per source, side by side (the four pair-solver files take 30-50 s each: 3 min in a row, 1 min in parallel),
    # then one link; objects under sigsvgd_amd/_obj/ (git-ignored)
    from concurrent.futures import ThreadPoolExecutor

    os.makedirs(objdir, exist_ok=True)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"] + [f"-D{d}" for d in defines]

    def compile_one(src):
        obj = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
        cmd = [_hipcc()] + flags + ["-c", os.path.join(_CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr}")
        return obj

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), max(1, (os.cpu_count() or 2) - 1))) as pool:
        objs = list(pool.map(compile_one, SOURCES))
    cmd = [_hipcc(), "--offload-arch=gfx950", "-fPIC", "-shared", "-o", lib_path] + objs + ["-ldl"]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    _check_dpp_hazards(lib_path)
    return lib_path


def _check_dpp_hazards(lib_path: str) -> None:
    """The kernels' inline-asm DPP moves/adds rely on hipcc's schedule for the 2 wait states after a VALU
    write of their source (csrc/gram_fast.hip); verify that on the disassembly of what was just built and
    refuse the library otherwise (scripts/check_dpp_hazards.py)."""
    import importlib.util

    script = os.path.join(_PKG, "..", "scripts", "check_dpp_hazards.py")
    if not os.path.exists(script):
        return
    spec = importlib.util.spec_from_file_location("check_dpp_hazards", script)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    total, bad = 0, []
    for text in mod.disassemble(lib_path):
        n, b = mod.check_disassembly(text)
        total += n
        bad += b
    if bad or total == 0:
        os.replace(lib_path, lib_path + ".rejected")
        raise RuntimeError(f"sigsvgd_amd: {len(bad)} DPP data hazards (of {total} DPP instructions) in the library "
                           f"hipcc produced, e.g. {bad[:2]}; kept as {lib_path}.rejected")


def load():
    """Load the library and declare signatures.  Raises RuntimeError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"sigsvgd_amd: HIP extension {LIB_PATH} is not built; run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). There is no CPU fallback."
        )
    # PyTorch-ROCm ships its own libamdhip64/libhsa-runtime64; they must be the ones already in the
    # process when this library's NEEDED entries are resolved, or two HIP runtimes end up loaded
    # (symptom: "no ROCm-capable device is detected" from the second one).
    import torch  # noqa: F401

    L = ctypes.CDLL(LIB_PATH)
    vp, ci, cd, cu, cf = ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_uint, ctypes.c_float
    L.sigsvgd_abi_version.restype = ci
    L.sigsvgd_abi_version.argtypes = []
    L.sigsvgd_last_error.restype = ctypes.c_char_p
    L.sigsvgd_last_error.argtypes = []
    L.sigsvgd_gram_workspace_bytes.restype = ci
    L.sigsvgd_gram_workspace_bytes.argtypes = [ci, ci, ci, ci, ci, ci, ci, cu, ctypes.POINTER(ctypes.c_size_t)]
    L.sigsvgd_gram_fwd.restype = ci
    L.sigsvgd_gram_fwd.argtypes = [vp, vp, ci, ci, ci, ci, ci, cd, ci, ci, cu, vp, vp, ctypes.c_size_t, vp]
    L.sigsvgd_gram_fwd_bwd.restype = ci
    L.sigsvgd_gram_fwd_bwd.argtypes = [vp, vp, ci, ci, ci, ci, ci, cd, ci, ci, cu, vp, vp, vp, vp, ctypes.c_size_t, vp]
    L.sigsvgd_gram_sym_partial.restype = ci
    L.sigsvgd_gram_sym_partial.argtypes = [vp, ci, ci, ci, ci, cd, ci, cu, ci, ci, vp, vp, vp, vp, ctypes.c_size_t, vp]
    L.sigsvgd_gram_sym_tile_rows.restype = ci
    L.sigsvgd_gram_sym_tile_rows.argtypes = [ci, ci]
    L.sigsvgd_svgd_phi.restype = ci
    L.sigsvgd_svgd_phi.argtypes = [vp, vp, vp, vp, ci, ci, vp, vp, vp, cf, vp]
    L.sigsvgd_svgd_step.restype = ci
    L.sigsvgd_svgd_step.argtypes = [vp, vp, vp, vp, ci, ci, vp, vp, vp, cf, vp, vp]
    L.sigsvgd_svgd_adam_step.restype = ci
    L.sigsvgd_svgd_adam_step.argtypes = [vp, vp, vp, vp, ci, ci, vp, vp, vp, cd, cd, cd, cd, vp, vp, vp, vp]
    L.sigsvgd_vec_sqdist.restype = ci
    L.sigsvgd_vec_sqdist.argtypes = [vp, vp, vp, vp, ci, ci, ci, ci, vp, vp]
    L.sigsvgd_vec_kernel.restype = ci
    L.sigsvgd_vec_kernel.argtypes = [vp, vp, vp, vp, ci, ci, ci, ci, ci, cd, cd, vp, vp, vp]
    L.sigsvgd_vec_kernel_fused.restype = ci
    L.sigsvgd_vec_kernel_fused.argtypes = [vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, cd, cd, vp, vp, vp, ctypes.c_size_t, vp]
    L.sigsvgd_vec_fused_workspace_bytes.restype = ci
    L.sigsvgd_vec_fused_workspace_bytes.argtypes = [ci, ci, ci, ctypes.POINTER(ctypes.c_size_t)]
    L.sigsvgd_obstacle_cost.restype = ci
    L.sigsvgd_obstacle_cost.argtypes = [vp, ci, ci, ci, vp, vp, vp, ci, vp, vp, vp, ci, cf, cf, vp, vp, vp, vp]
    L.sigsvgd_signature.restype = ci
    L.sigsvgd_signature.argtypes = [vp, ci, ci, ci, ci, ci, ci, vp, ctypes.POINTER(ctypes.c_longlong), vp]
    L.sigsvgd_signature_backward.restype = ci
    L.sigsvgd_signature_backward.argtypes = [vp, vp, ci, ci, ci, ci, ci, ci, vp, vp]
    if L.sigsvgd_abi_version() != ABI_VERSION:
        raise RuntimeError("sigsvgd_amd: libsigsvgd_hip.so ABI version mismatch; rebuild it")
    _lib = L
    return L


def last_error() -> str:
    return load().sigsvgd_last_error().decode("utf-8", "replace")


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError(f"sigsvgd_amd: {what} failed (rc={rc}): {last_error()}")
