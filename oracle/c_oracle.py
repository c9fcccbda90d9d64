"""ctypes loader for the C restatement (oracle/sigkernel_c.c).  Test infrastructure / cpu_baseline
only -- see the header of sigkernel_c.c.  `build()` compiles it with gcc (no reference sources are
involved; the reference has no native code for this path)."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "sigkernel_c.c")
_SO = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


def build(force: bool = False) -> str:
    os.makedirs(os.path.dirname(_SO), exist_ok=True)
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(_SRC):
        cmd = ["gcc", "-O3", "-fopenmp", "-shared", "-fPIC", "-o", _SO, _SRC, "-lm"]
        subprocess.run(cmd, check=True)
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = ctypes.CDLL(_SO)
        dp = ctypes.POINTER(ctypes.c_double)
        fp = ctypes.POINTER(ctypes.c_float)
        L.oracle_gram_fwd_bwd.restype = ctypes.c_int
        L.oracle_gram_fwd_bwd.argtypes = [fp, fp] + [ctypes.c_int] * 4 + [ctypes.c_double] + [ctypes.c_int] * 3 + [
            dp, ctypes.c_int, ctypes.c_int, dp, dp, ctypes.c_int]
        L.oracle_svgd_update.restype = ctypes.c_int
        L.oracle_svgd_update.argtypes = [dp, dp, dp, dp, ctypes.c_int, ctypes.c_int, ctypes.c_double, dp, dp]
        L.oracle_num_threads.restype = ctypes.c_int
        _lib = L
    return _lib


def _f32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def gram_fwd_bwd(X, Y, h=1.0, n=0, naive=False, kind=0, grad_out=None, rows=None, want_grad=True,
                 nthreads=0):
    """X [A,T,d], Y [B,T,d] (rounded to fp32 first, as the product path's I/O) ->
    (K [rows,B] f64, gradX [rows,T,d] f64 or None) for rows = (i0, i1) of X (default all)."""
    L = lib()
    X = _f32(X)
    Y = _f32(Y)
    A, T, d = X.shape
    B = Y.shape[0]
    i0, i1 = (0, A) if rows is None else rows
    K = np.empty((i1 - i0, B), dtype=np.float64)
    gX = np.empty((i1 - i0, T, d), dtype=np.float64) if want_grad else None
    dp = ctypes.POINTER(ctypes.c_double)
    fp = ctypes.POINTER(ctypes.c_float)
    go = None
    if grad_out is not None:
        go = np.ascontiguousarray(np.asarray(grad_out, dtype=np.float64))
        assert go.shape == (i1 - i0, B)
    rc = L.oracle_gram_fwd_bwd(
        X.ctypes.data_as(fp), Y.ctypes.data_as(fp), A, B, T, d, 1.0 / float(h), int(n), int(bool(naive)),
        int(kind), go.ctypes.data_as(dp) if go is not None else None, i0, i1, K.ctypes.data_as(dp),
        gX.ctypes.data_as(dp) if gX is not None else None, int(nthreads))
    if rc != 0:
        raise RuntimeError(f"oracle_gram_fwd_bwd failed rc={rc}")
    return K, gX


def svgd_update(K, score, grad_k, X, lr):
    L = lib()
    K = np.ascontiguousarray(K, dtype=np.float64)
    N = K.shape[0]
    s = np.ascontiguousarray(np.asarray(score, np.float64).reshape(N, -1))
    gk = np.ascontiguousarray(np.asarray(grad_k, np.float64).reshape(N, -1))
    Xf = np.ascontiguousarray(np.asarray(X, np.float64).reshape(N, -1))
    phi = np.empty_like(s)
    Xn = np.empty_like(s)
    dp = ctypes.POINTER(ctypes.c_double)
    L.oracle_svgd_update(K.ctypes.data_as(dp), s.ctypes.data_as(dp), gk.ctypes.data_as(dp), Xf.ctypes.data_as(dp),
                         N, s.shape[1], float(lr), phi.ctypes.data_as(dp), Xn.ctypes.data_as(dp))
    return phi.reshape(np.shape(score)), Xn.reshape(np.shape(X))


def num_threads() -> int:
    return int(lib().oracle_num_threads())
