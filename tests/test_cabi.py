"""CPU tests of the C-ABI shared library: it loads, exports every symbol include/*.h declares, and its
host-side argument checking / workspace queries work without a GPU (no compute calls here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    names = []
    inc = os.path.join(ROOT, "include")
    for fn in os.listdir(inc):
        if fn.endswith(".h"):
            txt = open(os.path.join(inc, fn)).read()
            txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
            names += re.findall(r"\b(sigsvgd_[a-z_0-9]+)\s*\(", txt)
    return sorted(set(names))


@pytest.fixture(scope="module")
def lib():
    from sigsvgd_amd import _lib

    if _lib.needs_build():
        _lib.build()
    return _lib.load()


def test_every_declared_symbol_is_exported(lib):
    from sigsvgd_amd import _lib

    declared = _declared_functions()
    assert len(declared) >= 7
    assert sorted(_lib.EXPORTS) == declared  # the Python binding covers exactly the header
    for name in declared:
        assert getattr(lib, name) is not None


def test_abi_version(lib):
    from sigsvgd_amd import _lib

    assert lib.sigsvgd_abi_version() == _lib.ABI_VERSION


def test_workspace_query_is_host_only(lib):
    n = ctypes.c_size_t(123)
    # register-resident path (n=0, T<=64), sized for the symmetric launch: one fp32 row of T*d per (8-row tile, column) item
    # of the upper triangle + one fp64 block of 8*T*d per (workgroup, tile) segment (at most tiles + workgroups of them)
    assert lib.sigsvgd_gram_workspace_bytes(1024, 1024, 64, 7, 0, 0, 1, 0, ctypes.byref(n)) == 0
    items = sum(1024 - 8 * k for k in range(128))
    lo = items * 448 * 4 + 128 * 8 * 448 * 8
    assert lo <= n.value <= lo + 1024 * 8 * 448 * 8 + 8192  # (the workgroup count depends on the device: <= 1024 here)
    # forward only: nothing is accumulated
    assert lib.sigsvgd_gram_workspace_bytes(1024, 1024, 64, 7, 0, 0, 0, 0, ctypes.byref(n)) == 0 and 0 < n.value <= 4096
    # paths in one to three channels: one byte per pair more, the flags of the pairs the coverage kernel solves again in fp64
    for d in (1, 2, 3, 4):
        assert lib.sigsvgd_gram_workspace_bytes(1024, 1024, 64, d, 0, 0, 0, 0, ctypes.byref(n)) == 0
        assert 1024 * 1024 <= n.value <= 1024 * 1024 + 4096
    # the static kernel decides the solver (ABI 9): the linear kernel runs every shape on the coverage kernel, whose scratch
    # is sized differently (ABI 8 sized a refined 9 x 7 launch at 70,912 B for a kernel that needs 5.8 MB)
    m = ctypes.c_size_t(0)
    assert lib.sigsvgd_gram_workspace_bytes(9, 7, 30, 3, 2, 0, 1, 0, ctypes.byref(n)) == 0
    assert lib.sigsvgd_gram_workspace_bytes(9, 7, 30, 3, 2, 1, 1, 0, ctypes.byref(m)) == 0
    assert lib.sigsvgd_gram_workspace_bytes(9, 7, 30, 3, 2, 0, 1, 8, ctypes.byref(n)) == 0  # (forced coverage kernel, RBF)
    assert m.value >= 1 << 20 and m.value == n.value
    assert lib.sigsvgd_gram_workspace_bytes(9, 7, 30, 3, 2, 5, 1, 0, ctypes.byref(m)) == -1  # unknown static kernel
    # generic path (dyadic refinement): partial slabs + per-workgroup forward-solution scratch
    assert lib.sigsvgd_gram_workspace_bytes(16, 16, 20, 2, 2, 0, 1, 0, ctypes.byref(n)) == 0 and n.value > 0
    # does not fit in LDS -> UNSUPPORTED with a message
    assert lib.sigsvgd_gram_workspace_bytes(4, 4, 400, 3, 0, 0, 1, 0, ctypes.byref(n)) == -2
    assert b"LDS" in lib.sigsvgd_last_error()
    assert lib.sigsvgd_gram_workspace_bytes(4, 4, 10, 3, 0, 0, 1, 0, None) == -1


def test_vec_fused_workspace_query(lib):
    """ABI 9: scratch of the reproducible route of the fused vector kernel = one [A][D] fp32 block per column split"""
    n = ctypes.c_size_t(0)
    assert lib.sigsvgd_vec_fused_workspace_bytes(1024, 1024, 448, ctypes.byref(n)) == 0
    assert n.value > 0 and n.value % (1024 * 448 * 4) == 0 and n.value // (1024 * 448 * 4) <= 16  # (16 column tiles of 64)
    assert lib.sigsvgd_vec_fused_workspace_bytes(64, 64, 7, ctypes.byref(n)) == 0 and n.value == 0  # one split: nothing to join
    assert lib.sigsvgd_vec_fused_workspace_bytes(0, 64, 7, ctypes.byref(n)) == -1
    assert lib.sigsvgd_vec_fused_workspace_bytes(64, 64, 7, None) == -1


def test_workspace_query_covers_every_pair_kernel(lib):
    """one query per kernel family (register-resident, quadrant, refined-grid, band, coverage), gradient and forward-only:
    status 0 and a size; the kernels that flag cancelled pairs for the fp64 pass need A * B bytes of flags even forward-only"""
    n = ctypes.c_size_t(0)
    shapes = {"fast": (64, 7, 0), "fast32": (32, 7, 0), "quad": (128, 14, 0), "dyad": (20, 2, 2), "dyad5": (5, 2, 5),
              "band notebook": (10, 2, 4), "band maze": (30, 2, 3), "coverage": (40, 3, 3), "coverage naive": (20, 2, 2)}
    for name, (T, d, order) in shapes.items():
        flags = 1 if name.endswith("naive") else 0
        for want_grad in (0, 1):
            for A, B in ((37, 37), (5, 9)):
                assert lib.sigsvgd_gram_workspace_bytes(A, B, T, d, order, 0, want_grad, flags, ctypes.byref(n)) == 0, name
                assert n.value > 0, name
                if not name.startswith("coverage") and not name.startswith("fast"):
                    assert n.value >= A * B
    # the same shapes forced onto the coverage kernel
    for name, (T, d, order) in shapes.items():
        assert lib.sigsvgd_gram_workspace_bytes(9, 9, T, d, order, 0, 1, 8, ctypes.byref(n)) == 0 and n.value > 0, name


def test_argument_errors_are_status_codes(lib):
    """bad arguments are rejected on the host before anything is launched"""
    one = ctypes.c_void_p(16)  # never dereferenced: the checks below fail first
    rc = lib.sigsvgd_gram_fwd(None, one, 2, 2, 5, 2, 0, 1.0, 0, 0, 0, one, None, 0, None)
    assert rc == -1 and b"null" in lib.sigsvgd_last_error()
    rc = lib.sigsvgd_gram_fwd(one, one, 2, 2, 1, 2, 0, 1.0, 0, 0, 0, one, None, 0, None)
    assert rc == -1  # T < 2
    rc = lib.sigsvgd_gram_fwd(one, one, 2, 2, 5, 2, 7, 1.0, 0, 0, 0, one, None, 0, None)
    assert rc == -1 and b"dtype" in lib.sigsvgd_last_error()
    rc = lib.sigsvgd_gram_fwd(one, one, 2, 2, 5, 2, 0, -1.0, 0, 0, 0, one, None, 0, None)
    assert rc == -1 and b"inv_h" in lib.sigsvgd_last_error()
    rc = lib.sigsvgd_gram_fwd_bwd(one, one, 2, 2, 5, 2, 0, 1.0, 0, 0, 0, None, one, None, None, 0, None)
    assert rc == -1 and b"gradX_out" in lib.sigsvgd_last_error()
    rc = lib.sigsvgd_gram_fwd_bwd(one, one, 2, 3, 5, 2, 0, 1.0, 0, 0, 4, None, one, one, None, 0, None)
    assert rc == -1  # Y_IS_X with A != B
    rc = lib.sigsvgd_gram_fwd_bwd(one, one, 2, 2, 5, 2, 0, 1.0, 0, 0, 0, None, one, one, None, 0, None)
    assert rc == -3 and b"workspace" in lib.sigsvgd_last_error()  # workspace missing
    rc = lib.sigsvgd_gram_sym_partial(one, 4, 5, 2, 0, 1.0, 0, 0, 2, 2, None, one, one, None, 0, None)
    assert rc == -1  # tile_offset >= tile_stride
    rc = lib.sigsvgd_gram_sym_partial(one, 4, 200, 2, 0, 1.0, 0, 0, 0, 1, None, one, one, None, 0, None)
    assert rc == -2  # T > 128 is outside the register-resident and streaming paths
    rc = lib.sigsvgd_svgd_phi(None, one, one, None, 4, 4, one, None, None, 0.1, None)
    assert rc == -1


def test_build_from_a_clean_tree(tmp_path):
    """`build()` proves itself: the sources alone, compiled into an empty directory (no object or library of an earlier
    build is touched), give a library that exports exactly the header's entry points, reports the header's ABI version and
    passes the DPP hazard check.  (The driver's build() call is mtime-gated and usually finds the in-tree library current.)"""
    import re
    import subprocess

    from sigsvgd_amd import _lib

    out = tmp_path / "libsigsvgd_clean.so"
    _lib.build(out_path=str(out), objdir=str(tmp_path / "obj"))
    assert out.exists() and out.stat().st_size > 1 << 20
    syms = subprocess.run(["nm", "-D", "--defined-only", str(out)], capture_output=True, text=True, check=True).stdout
    exported = sorted(set(re.findall(r"\bT (sigsvgd_\w+)", syms)))
    assert exported == sorted(_lib.EXPORTS)
    header = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "sigsvgd_hip.h")).read()
    abi = int(re.search(r"#define SIGSVGD_ABI_VERSION (\d+)", header).group(1))
    L = ctypes.CDLL(str(out))
    L.sigsvgd_abi_version.restype = ctypes.c_int
    assert L.sigsvgd_abi_version() == abi == _lib.ABI_VERSION
