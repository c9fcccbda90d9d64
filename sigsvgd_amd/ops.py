"""Thin torch-facing wrappers over the C ABI (the ONLY place the package touches the HIP library).

Every function takes/returns torch tensors living on an MI355X (`device.type == "cuda"` on ROCm),
checks dtype/contiguity, passes raw device pointers + the current HIP stream to the library and
converts status codes into RuntimeError.  CPU tensors are rejected: there is no CPU fallback.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Tuple

import torch

from . import _lib

_WS = {}  # (device index, stream) -> uint8 workspace tensor (never needs zeroing: partial sums are stored, not accumulated)


def _require_gpu(*tensors) -> torch.device:
    dev = None
    for t in tensors:
        if t is None:
            continue
        if t.device.type != "cuda":
            raise RuntimeError(
                "sigsvgd_amd: the signature-kernel/SVGD hot path runs only on a HIP device (MI355X); "
                f"got a tensor on '{t.device}'. There is no CPU fallback."
            )
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError(f"sigsvgd_amd: tensors on different devices ({dev} vs {t.device})")
    return dev


def _stream_ptr(dev: torch.device) -> ctypes.c_void_p:
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _workspace(dev: torch.device, nbytes: int) -> Tuple[Optional[torch.Tensor], int]:
    if nbytes == 0:
        return None, 0
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
    ws = _WS.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8, device=dev)
        _WS[key] = ws
    return ws, ws.numel()


def _io_dtype(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return _lib.F32
    if t.dtype == torch.float64:
        return _lib.F64
    raise TypeError(f"sigsvgd_amd: paths must be float32 or float64, got {t.dtype}")


def pad_to_length(P: torch.Tensor, T: int) -> torch.Tensor:
    """[batch, t, d] -> [batch, T, d] with the LAST point repeated.  Exact for the signature kernel: the repeated points add
    zero increments, for which the Goursat stencil copies the solution along the added rows / columns (in every kernel of
    the library: the 4-corner increment of two equal rows of the static kernel is an exact zero), at any dyadic order."""
    t = P.shape[1]
    if t == T:
        return P
    return torch.cat([P, P[:, -1:, :].expand(-1, T - t, -1)], dim=1)


def fold_padded_grad(g: torch.Tensor, t: int) -> torch.Tensor:
    """gradient w.r.t. a path padded by pad_to_length -> gradient w.r.t. the path itself: the copies of the last point add up"""
    if g.shape[1] == t:
        return g
    out = g[:, :t].clone()
    out[:, t - 1] += g[:, t:].sum(dim=1)
    return out


def _prep_paths(X: torch.Tensor, Y: torch.Tensor):
    """-> (X, Y) contiguous, detached, of one dtype and ONE length: upstream sigkernel takes paths of different lengths
    (no caller in the reference does, src/kernels/_traj_kernels.py:200); the shorter batch is padded with its last point,
    which leaves every K[i, j] unchanged (pad_to_length).  The callers fold the gradient back (fold_padded_grad)."""
    if X.dim() != 3 or Y.dim() != 3:
        raise ValueError(f"paths must be [batch, length, dim]; got {tuple(X.shape)} and {tuple(Y.shape)}")
    if X.shape[2] != Y.shape[2]:
        raise ValueError(f"X and Y must share the path dimension (got {tuple(X.shape)} vs {tuple(Y.shape)})")
    if X.dtype != Y.dtype:
        Y = Y.to(X.dtype)
    if X.shape[0] == 0 or Y.shape[0] == 0:
        raise ValueError("empty batch")
    if X.shape[1] < 2 or Y.shape[1] < 2:
        raise ValueError("paths need at least 2 points")
    T = max(X.shape[1], Y.shape[1])
    return pad_to_length(X.detach(), T).contiguous(), pad_to_length(Y.detach(), T).contiguous()


def _flags(naive: bool, sym: bool, y_is_x: bool, force_generic: bool, stored_forward: bool = False) -> int:
    f = 0
    if naive:
        f |= _lib.FLAG_NAIVE_SOLVER
    if sym:
        f |= _lib.FLAG_SYM
    if y_is_x:
        f |= _lib.FLAG_Y_IS_X
    if force_generic:
        f |= _lib.FLAG_FORCE_GENERIC
    if stored_forward:
        f |= _lib.FLAG_STORED_FORWARD
    return f


def gram_fwd(X, Y, inv_h: float, dyadic_order: int = 0, static_kind: int = _lib.STATIC_RBF,
             naive: bool = False, force_generic: bool = False, y_is_x: bool = False,
             stored_forward: bool = False) -> torch.Tensor:
    """K[A,B] = signature-kernel Gram matrix (forward only).  y_is_x: the caller states that Y holds the
    same values as X, so each unordered pair is solved once and K is mirrored."""
    L = _lib.load()
    dev = _require_gpu(X, Y)
    Xc, Yc = _prep_paths(X, Y)
    A, T, d = Xc.shape
    B = Yc.shape[0]
    if y_is_x and X.shape[1] != Y.shape[1]:
        raise ValueError("y_is_x needs X and Y of one shape")
    flags = _flags(naive, False, bool(y_is_x) and A == B, force_generic, stored_forward)
    nbytes = ctypes.c_size_t(0)
    _lib.check(L.sigsvgd_gram_workspace_bytes(A, B, T, d, dyadic_order, int(static_kind), 0, flags, ctypes.byref(nbytes)),
               "gram_workspace_bytes")
    ws, wsn = _workspace(dev, nbytes.value)
    K = torch.empty((A, B), dtype=Xc.dtype, device=dev)
    with torch.cuda.device(dev):
        rc = L.sigsvgd_gram_fwd(Xc.data_ptr(), Yc.data_ptr(), A, B, T, d, _io_dtype(Xc), float(inv_h),
                                int(dyadic_order), int(static_kind), flags, K.data_ptr(),
                                ws.data_ptr() if ws is not None else None, wsn, _stream_ptr(dev))
    _lib.check(rc, "gram_fwd")
    return K


def gram_fwd_bwd(X, Y, inv_h: float, dyadic_order: int = 0, static_kind: int = _lib.STATIC_RBF,
                 grad_out: Optional[torch.Tensor] = None, naive: bool = False, sym: bool = False,
                 y_is_x: bool = False, force_generic: bool = False,
                 check_regime: bool = True, stored_forward: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """(K[A,B], gradX[A,T,d]) with gradX = d sum(grad_out*K)/dX (first slot); grad_out None = ones.

    Bit-reproducible: every reduction over pairs runs in an order fixed by the launch geometry (no floating-point
    atomics), so two calls on the same input -- eager or replayed from a captured graph -- return the same bits.

    Accuracy (include/sigsvgd_hip.h): every entry of K within 1e-5 of the fp64 reference's, relative to max(|K|, 0.1) --
    the fp32-sweep kernels check every pair for cancellation and for its conditioning in the increments and hand the pairs
    that fail to an exact fp64 pass inside the same call; the gradient of such a pair keeps the fp32 solution (its error is
    relative to the largest gradient entry of the launch).  `force_generic=True`: fp64 sweeps for K and the gradient alike.
    `check_regime` and `stored_forward` are accepted for callers written against earlier versions (when long paths
    could run on a kernel that regenerated the forward solution and declined rough pairs) and have no effect."""
    L = _lib.load()
    dev = _require_gpu(X, Y, grad_out)
    Xc, Yc = _prep_paths(X, Y)
    A, T, d = Xc.shape
    B = Yc.shape[0]
    go = None
    if grad_out is not None:
        if tuple(grad_out.shape) != (A, B):
            raise ValueError(f"grad_out must be [{A},{B}], got {tuple(grad_out.shape)}")
        go = grad_out.detach().to(Xc.dtype).contiguous()
    if (y_is_x or sym) and X.shape[1] != Y.shape[1]:
        raise ValueError("y_is_x / sym need X and Y of one shape")
    flags = _flags(naive, sym, y_is_x, force_generic, stored_forward)
    nbytes = ctypes.c_size_t(0)
    _lib.check(L.sigsvgd_gram_workspace_bytes(A, B, T, d, dyadic_order, int(static_kind), 1, flags, ctypes.byref(nbytes)),
               "gram_workspace_bytes")
    ws, wsn = _workspace(dev, nbytes.value)
    K = torch.empty((A, B), dtype=Xc.dtype, device=dev)
    gX = torch.empty((A, T, d), dtype=Xc.dtype, device=dev)
    with torch.cuda.device(dev):
        rc = L.sigsvgd_gram_fwd_bwd(Xc.data_ptr(), Yc.data_ptr(), A, B, T, d, _io_dtype(Xc), float(inv_h),
                                    int(dyadic_order), int(static_kind), flags,
                                    go.data_ptr() if go is not None else None, K.data_ptr(), gX.data_ptr(),
                                    ws.data_ptr() if ws is not None else None, wsn, _stream_ptr(dev))
    _lib.check(rc, "gram_fwd_bwd")
    return K, fold_padded_grad(gX, X.shape[1])


def svgd_phi(K, score, grad_k, mask=None, X=None, lr: Optional[float] = None, adagrad_state=None,
             inplace: bool = False):
    """v = -((K @ score - grad_k)/N) [* mask]; with X and lr also returns X - lr*v.

    All fp32, shapes K [N,N], score/grad_k/mask/X [N, ...] (flattened to [N,D]).
    adagrad_state: contiguous fp32 tensor shaped like score, updated IN PLACE (state += v^2) and applied
    (v / sqrt(state + 1e-12)) before the update -- the reference's adaptive_gradient=True (svgd.py:110-113).
    Returns v (shaped like score) or (v, X_new)."""
    L = _lib.load()
    dev = _require_gpu(K, score, grad_k, mask, X)
    N = K.shape[0]
    if K.dim() != 2 or K.shape[1] != N:
        raise ValueError(f"K must be square, got {tuple(K.shape)}")
    shape = score.shape
    f = lambda t: t.detach().to(torch.float32).reshape(N, -1).contiguous()
    Kc = K.detach().to(torch.float32).contiguous()
    s, gk = f(score), f(grad_k)
    D = s.shape[1]
    if gk.shape != s.shape:
        raise ValueError(f"grad_k shape {tuple(grad_k.shape)} does not match score {tuple(score.shape)}")
    m = None
    if mask is not None:
        m = torch.broadcast_to(torch.as_tensor(mask, dtype=torch.float32, device=dev), shape)
        m = m.reshape(N, -1).contiguous()
    v = torch.empty_like(s)
    Xc = Xn = None
    if X is not None:
        if lr is None:
            raise ValueError("lr is required with X")
        Xc = f(X)
        if inplace:  # every element is read and written by the same thread
            if Xc.data_ptr() != X.data_ptr():
                raise ValueError("inplace update needs contiguous float32 particles")
            Xn = Xc
        else:
            Xn = torch.empty_like(Xc)
    ag = None
    if adagrad_state is not None:
        _require_gpu(adagrad_state)
        if adagrad_state.dtype != torch.float32 or not adagrad_state.is_contiguous() or adagrad_state.numel() != N * D:
            raise ValueError("adagrad_state must be a contiguous float32 tensor with the shape of score")
        ag = adagrad_state
    with torch.cuda.device(dev):
        rc = L.sigsvgd_svgd_step(Kc.data_ptr(), s.data_ptr(), gk.data_ptr(), m.data_ptr() if m is not None else None,
                                 N, D, v.data_ptr(), Xc.data_ptr() if Xc is not None else None,
                                 Xn.data_ptr() if Xn is not None else None, float(lr or 0.0),
                                 ag.data_ptr() if ag is not None else None, _stream_ptr(dev))
    _lib.check(rc, "svgd_step")
    v = v.reshape(shape)
    if X is not None:
        return v, (X if inplace else Xn.reshape(X.shape))
    return v


class AdamState:
    """State of the fused Adam update (torch.optim.Adam semantics): exp_avg / exp_avg_sq [N, D] fp32 and the step
    counter, all on the device (the counter too, so a captured graph can replay the update)."""

    def __init__(self, like: torch.Tensor, betas=(0.9, 0.999), eps: float = 1e-8):
        n = like.shape[0]
        self.exp_avg = torch.zeros((n, like.numel() // n), dtype=torch.float32, device=like.device)
        self.exp_avg_sq = torch.zeros_like(self.exp_avg)
        self.step = torch.zeros((), dtype=torch.int32, device=like.device)
        self.t_host = 0  # host mirror of the counter (no read-back needed to export the state)
        self.betas, self.eps = (float(betas[0]), float(betas[1])), float(eps)


def svgd_adam(K, score, grad_k, X, lr: float, state: AdamState, mask=None, inplace: bool = False):
    """v = -((K @ score - grad_k)/N) [* mask] and torch.optim.Adam's update of X along it, one launch (plus a
    one-thread launch that advances the device-side step counter).  Returns (v shaped like score, X_new);
    inplace=True writes the update into X itself (X must be contiguous fp32) and returns X."""
    L = _lib.load()
    dev = _require_gpu(K, score, grad_k, mask, X, state.exp_avg)
    N = K.shape[0]
    if K.dim() != 2 or K.shape[1] != N:
        raise ValueError(f"K must be square, got {tuple(K.shape)}")
    shape = score.shape
    f = lambda t: t.detach().to(torch.float32).reshape(N, -1).contiguous()
    Kc = K.detach().to(torch.float32).contiguous()
    s, gk, Xc = f(score), f(grad_k), f(X)
    D = s.shape[1]
    if gk.shape != s.shape or Xc.shape != s.shape or tuple(state.exp_avg.shape) != (N, D):
        raise ValueError("score, grad_k, X and the Adam state must share the shape [N, D]")
    m = None
    if mask is not None:
        m = torch.broadcast_to(torch.as_tensor(mask, dtype=torch.float32, device=dev), shape).reshape(N, -1).contiguous()
    v = torch.empty_like(s)
    if inplace:
        if Xc.data_ptr() != X.data_ptr():
            raise ValueError("inplace Adam update needs contiguous float32 particles")
        Xn = Xc  # every element is read and written by the same thread
    else:
        Xn = torch.empty_like(Xc)
    with torch.cuda.device(dev):
        rc = L.sigsvgd_svgd_adam_step(Kc.data_ptr(), s.data_ptr(), gk.data_ptr(), m.data_ptr() if m is not None else None,
                                      N, D, v.data_ptr(), Xc.data_ptr(), Xn.data_ptr(), float(lr), state.betas[0],
                                      state.betas[1], state.eps, state.exp_avg.data_ptr(), state.exp_avg_sq.data_ptr(),
                                      state.step.data_ptr(), _stream_ptr(dev))
    _lib.check(rc, "svgd_adam_step")
    state.t_host += 1
    return v.reshape(shape), (X if inplace else Xn.reshape(X.shape))


def gram_sym_partial(X, inv_h: float, tile_offset: int, tile_stride: int, static_kind: int = _lib.STATIC_RBF,
                     grad_out: Optional[torch.Tensor] = None, sym: bool = False, out=None, fold: bool = False):
    """This rank's share of the symmetric Gram + gradient on the gathered particles X [N,T,d]:
    returns (K_partial [N,N] X.dtype, grad_partial [N,T,d] fp64), zero outside the owned pairs.
    Summed over tile_offset = 0..tile_stride-1 they equal gram_fwd_bwd(X, X, y_is_x=True).
    The launch owns the row tiles (`sym_tile_rows(T, d)` rows each) tile_offset + k*tile_stride; with fold=True also
    their mirror images, which gives every rank the same number of pairs (SIGSVGD_FLAG_FOLD_TILES).
    `out=(K_partial, grad_partial)` reuses the caller's buffers (K_partial is re-zeroed, grad_partial overwritten)."""
    L = _lib.load()
    dev = _require_gpu(X, grad_out)
    Xc, _ = _prep_paths(X, X)
    N, T, d = Xc.shape
    go = None
    if grad_out is not None:
        go = grad_out.detach().to(Xc.dtype).contiguous()
    if out is not None:
        Kp, gp = out
        if (tuple(Kp.shape) != (N, N) or Kp.dtype != Xc.dtype or not Kp.is_contiguous() or tuple(gp.shape) != (N, T, d)
                or gp.dtype != torch.float64 or not gp.is_contiguous()):
            raise ValueError("out must be (K_partial [N,N] of X's dtype, grad_partial [N,T,d] float64), contiguous")
        Kp.zero_()
    else:
        Kp = torch.zeros((N, N), dtype=Xc.dtype, device=dev)
        gp = torch.empty((N, T, d), dtype=torch.float64, device=dev)  # fully overwritten by the library
    flags = _flags(False, sym, True, False) | (_lib.FLAG_FOLD_TILES if fold else 0)
    nbytes = ctypes.c_size_t(0)
    _lib.check(L.sigsvgd_gram_workspace_bytes(N, N, T, d, 0, int(static_kind), 1, flags, ctypes.byref(nbytes)), "gram_workspace_bytes")
    ws, wsn = _workspace(dev, nbytes.value)
    with torch.cuda.device(dev):
        rc = L.sigsvgd_gram_sym_partial(Xc.data_ptr(), N, T, d, _io_dtype(Xc), float(inv_h), int(static_kind),
                                        flags, int(tile_offset), int(tile_stride),
                                        go.data_ptr() if go is not None else None, Kp.data_ptr(), gp.data_ptr(),
                                        ws.data_ptr() if ws is not None else None, wsn, _stream_ptr(dev))
    _lib.check(rc, "gram_sym_partial")
    return Kp, gp


def sym_tile_rows(T: int, d: int) -> int:
    """Rows per tile of the symmetric / partial solve (the ownership unit of the sharded step); 0 for shapes the partial
    solve does not take.  Host-only query of the library."""
    try:
        return int(_lib.load().sigsvgd_gram_sym_tile_rows(int(T), int(d)))
    except RuntimeError:
        # Host-only rule, restated for boxes without the built library (the CPU test doubles of tests/helpers.py and the gloo
        # rehearsal use it to mirror the ownership; every compute entry point still raises without the library):
        # csrc/gram_fast.hip grad_nw (T <= 64: 8 rows, 4 with d > 8), csrc/gram_quad.hip (65 <= T <= 128: 8 rows)
        T, d = int(T), int(d)
        if 3 <= T <= 64 and d <= 16:
            return 8 if d <= 8 else 4
        if 65 <= T <= 128 and d <= 16:
            return 8
        return 0


def owned_tiles(ntile: int, tile_offset: int, tile_stride: int, fold: bool = False):
    """The row tiles `gram_sym_partial(..., tile_offset, tile_stride, fold=)` owns, in the library's order (mirror of
    csrc/sig_common.h TileMap; used by the sharding tests and the CPU test doubles)."""
    first = [t for t in range(tile_offset, ntile, tile_stride) if not fold or t <= (ntile - 1) // 2]
    second = [ntile - 1 - t for t in range(tile_offset, ntile, tile_stride) if 2 * t < ntile - 1] if fold else []
    return first + second


# ---- vector kernels / truncated signature ---------------------------------------
def _prep_vec(t: torch.Tensor, dtype=None) -> torch.Tensor:
    t = t.detach()
    if t.dim() < 2:
        t = torch.atleast_2d(t)
    t = t.flatten(1)
    if dtype is not None and t.dtype != dtype:
        t = t.to(dtype)
    return t.contiguous()


def vec_sqdist(X, Y, XM=None, YM=None) -> torch.Tensor:
    """sq[A,B] = clamp(sum_c (XM - YM)_c (X - Y)_c, 0); XM = YM = None: |x_i - y_j|^2.  X [A,D], Y [B,D]."""
    L = _lib.load()
    dev = _require_gpu(X, Y, XM, YM)
    Xc = _prep_vec(X)
    dt = _io_dtype(Xc)
    Yc = _prep_vec(Y, Xc.dtype)
    if Xc.shape[1] != Yc.shape[1]:
        raise ValueError(f"X and Y must share the feature size, got {tuple(Xc.shape)} vs {tuple(Yc.shape)}")
    if (XM is None) != (YM is None):
        raise ValueError("XM and YM must both be given or both be None")
    XMc = YMc = None
    if XM is not None:
        XMc, YMc = _prep_vec(XM, Xc.dtype), _prep_vec(YM, Xc.dtype)
        if XMc.shape != Xc.shape or YMc.shape != Yc.shape:
            raise ValueError("XM / YM must have the shapes of X / Y")
    A, D = Xc.shape
    B = Yc.shape[0]
    if A == 0 or B == 0 or D == 0:
        raise ValueError("empty batch")
    sq = torch.empty((A, B), dtype=Xc.dtype, device=dev)
    with torch.cuda.device(dev):
        rc = L.sigsvgd_vec_sqdist(Xc.data_ptr(), Yc.data_ptr(), XMc.data_ptr() if XMc is not None else None,
                                  YMc.data_ptr() if YMc is not None else None, A, B, D, dt, sq.data_ptr(),
                                  _stream_ptr(dev))
    _lib.check(rc, "vec_sqdist")
    return sq


def vec_kernel(sq, XM, YM, kind: int, inv_h2: float, grad_scale: float, grad_out=None, want_K: bool = True,
               want_grad: bool = True):
    """(K[A,B] or None, dK[A,D] or None): K = f(sq), dK = grad_scale * sum_j grad_out_ij w(sq_ij) (XM_i - YM_j)."""
    L = _lib.load()
    dev = _require_gpu(sq, XM, YM, grad_out)
    sqc = sq.detach().contiguous()
    dt = _io_dtype(sqc)
    A, B = sqc.shape
    XMc = YMc = go = None
    D = 1
    if want_grad:
        XMc, YMc = _prep_vec(XM, sqc.dtype), _prep_vec(YM, sqc.dtype)
        D = XMc.shape[1]
        if XMc.shape[0] != A or YMc.shape != (B, D):
            raise ValueError(f"XM {tuple(XMc.shape)} / YM {tuple(YMc.shape)} do not match sq {tuple(sqc.shape)}")
    if grad_out is not None:
        if tuple(grad_out.shape) != (A, B):
            raise ValueError(f"grad_out must be [{A},{B}], got {tuple(grad_out.shape)}")
        go = grad_out.detach().to(sqc.dtype).contiguous()
    K = torch.empty((A, B), dtype=sqc.dtype, device=dev) if want_K else None
    dK = torch.empty((A, D), dtype=sqc.dtype, device=dev) if want_grad else None
    with torch.cuda.device(dev):
        rc = L.sigsvgd_vec_kernel(sqc.data_ptr(), XMc.data_ptr() if XMc is not None else None,
                                  YMc.data_ptr() if YMc is not None else None,
                                  go.data_ptr() if go is not None else None, A, B, D, dt, int(kind), float(inv_h2),
                                  float(grad_scale), K.data_ptr() if K is not None else None,
                                  dK.data_ptr() if dK is not None else None, _stream_ptr(dev))
    _lib.check(rc, "vec_kernel")
    return K, dK


def vec_fused_supported(X: torch.Tensor) -> bool:
    """Shapes / dtypes `vec_kernel_fused` takes (include/sigsvgd_hip.h): fp32, up to 512 channels."""
    return X.dtype == torch.float32 and 1 <= X.reshape(X.shape[0], -1).shape[1] <= 512


def vec_kernel_fused(X, Y, kind: int, inv_h2: float, grad_scale: float, XM=None, YM=None, grad_out=None,
                     want_K: bool = True, want_grad: bool = True, reproducible: bool = True):
    """(K[A,B] or None, dK[A,D] or None) for a GIVEN bandwidth in one launch (`sigsvgd_vec_kernel_fused`): the
    distance never goes to HBM and both GEMM-shaped sums run on the fp32 matrix cores.  XM / YM = X M / Y M for the
    scaled kernels (both or neither).  reproducible (default): the column splits of the launch store their partial sums
    in a workspace and a second small launch adds them in a fixed order -- two calls return the same bits; False: one
    launch, the splits meet in fp32 atomics."""
    L = _lib.load()
    dev = _require_gpu(X, Y, XM, YM, grad_out)
    Xc, Yc = _prep_vec(X, torch.float32), _prep_vec(Y, torch.float32)
    A, D = Xc.shape
    B = Yc.shape[0]
    if Yc.shape[1] != D:
        raise ValueError(f"X {tuple(Xc.shape)} / Y {tuple(Yc.shape)} channel mismatch")
    if (XM is None) != (YM is None):
        raise ValueError("XM and YM must both be given or both be None")
    XMc = YMc = go = None
    if XM is not None:
        XMc, YMc = _prep_vec(XM, torch.float32), _prep_vec(YM, torch.float32)
        if XMc.shape != Xc.shape or YMc.shape != Yc.shape:
            raise ValueError(f"XM {tuple(XMc.shape)} / YM {tuple(YMc.shape)} do not match X / Y")
    if grad_out is not None:
        if tuple(grad_out.shape) != (A, B):
            raise ValueError(f"grad_out must be [{A},{B}], got {tuple(grad_out.shape)}")
        go = grad_out.detach().to(torch.float32).contiguous()
    K = torch.empty((A, B), dtype=torch.float32, device=dev) if want_K else None
    dK = torch.empty((A, D), dtype=torch.float32, device=dev) if want_grad else None
    p = lambda t: t.data_ptr() if t is not None else None
    ws, wsn = None, 0
    if want_grad and reproducible:  # per-split partial sums joined in a fixed order (the header's reproducible route)
        nbytes = ctypes.c_size_t(0)
        _lib.check(L.sigsvgd_vec_fused_workspace_bytes(A, B, D, ctypes.byref(nbytes)), "vec_fused_workspace_bytes")
        ws, wsn = _workspace(dev, nbytes.value)
    with torch.cuda.device(dev):
        rc = L.sigsvgd_vec_kernel_fused(Xc.data_ptr(), Yc.data_ptr(), p(XMc), p(YMc), p(go), A, B, D, _lib.F32, int(kind),
                                        float(inv_h2), float(grad_scale), p(K), p(dK), p(ws), wsn, _stream_ptr(dev))
    _lib.check(rc, "vec_kernel_fused")
    return K, dK


def signature_channels(channels: int, depth: int) -> int:
    L = _lib.load()
    n = ctypes.c_longlong(0)
    _lib.check(L.sigsvgd_signature(None, 1, 1, int(channels), int(depth), 0, _lib.F32, None, ctypes.byref(n), None),
               "signature (channel query)")
    return int(n.value)


def _signature_fwd(Xc: torch.Tensor, depth: int, basepoint: bool) -> torch.Tensor:
    L = _lib.load()
    dev = Xc.device
    dt = _io_dtype(Xc)
    N, Ln, C = Xc.shape
    n = ctypes.c_longlong(0)
    _lib.check(L.sigsvgd_signature(None, N, Ln, C, int(depth), int(bool(basepoint)), dt, None, ctypes.byref(n), None),
               "signature (channel query)")
    out = torch.empty((N, int(n.value)), dtype=Xc.dtype, device=dev)
    with torch.cuda.device(dev):
        rc = L.sigsvgd_signature(Xc.data_ptr(), N, Ln, C, int(depth), int(bool(basepoint)), dt, out.data_ptr(), None,
                                 _stream_ptr(dev))
    _lib.check(rc, "signature")
    return out


def signature_backward(X, grad_sig, depth: int, basepoint: bool = False) -> torch.Tensor:
    """d sum(grad_sig * signature(X, depth, basepoint)) / dX  -> [N, L, C] (`sigsvgd_signature_backward`)."""
    L = _lib.load()
    dev = _require_gpu(X, grad_sig)
    Xc = X.detach().contiguous()
    dt = _io_dtype(Xc)
    N, Ln, C = Xc.shape
    g = grad_sig.detach().to(Xc.dtype).contiguous()
    if g.dim() != 2 or g.shape[0] != N or g.shape[1] != signature_channels(C, depth):
        raise ValueError(f"grad_sig must be [{N}, {signature_channels(C, depth)}], got {tuple(g.shape)}")
    gX = torch.empty_like(Xc)
    with torch.cuda.device(dev):
        rc = L.sigsvgd_signature_backward(Xc.data_ptr(), g.data_ptr(), N, Ln, C, int(depth), int(bool(basepoint)), dt,
                                          gX.data_ptr(), _stream_ptr(dev))
    _lib.check(rc, "signature_backward")
    return gX


class _Signature(torch.autograd.Function):
    """signature(X) as an autograd node: HIP forward, HIP adjoint (the path is the only differentiable input)."""

    @staticmethod
    def forward(ctx, X, depth, basepoint):
        Xc = X.detach().contiguous()
        ctx.save_for_backward(Xc)
        ctx.depth, ctx.basepoint = int(depth), bool(basepoint)
        return _signature_fwd(Xc, depth, basepoint)

    @staticmethod
    def backward(ctx, grad_sig):
        (Xc,) = ctx.saved_tensors
        return signature_backward(Xc, grad_sig, ctx.depth, ctx.basepoint), None, None


def signature(X, depth: int, basepoint: bool = False) -> torch.Tensor:
    """Truncated signature of paths X [N, L, C] -> [N, C + ... + C^depth] (signatory's layout).  Differentiable with
    respect to X (reference: `signatory.signature` inside PathSigKernel, src/kernels/_traj_kernels.py:124-125, reached by
    autograd from src/inference/score.py:50-55)."""
    _lib.load()
    _require_gpu(X)
    if X.dim() != 3:
        raise ValueError(f"paths must be [batch, length, channels]; got {tuple(X.shape)}")
    if X.shape[0] == 0:
        raise ValueError("empty batch")
    _io_dtype(X)
    if X.requires_grad and torch.is_grad_enabled():
        return _Signature.apply(X, int(depth), bool(basepoint))
    return _signature_fwd(X.detach().contiguous(), depth, basepoint)


def obstacle_cost(x, start, target, basis, log_weights, mean, std, w_obstacle: float = 1.0, w_length: float = 1.0,
                  want_traj: bool = True, want_grad: bool = True):
    """Planning cost of the reference's obstacle-field script on the device with its analytic gradient
    (`sigsvgd_obstacle_cost`; examples/script_planning_obstacle_field.py:113-126).

    x [N, knots, d] interior knots, start/target [d], basis [samples, knots + 2], mixture log_weights [M]
    (normalised), mean/std [M, d].  Returns (cost [N], traj [N, samples, d] or None, d cost / d x or None)."""
    L = _lib.load()
    dev = _require_gpu(x, basis, mean, std, log_weights, start, target)
    if x.dim() != 3:
        raise ValueError(f"knots must be [batch, knots, channels]; got {tuple(x.shape)}")
    N, Kx, d = x.shape
    if N == 0:
        raise ValueError("empty batch")
    f = lambda t: t.detach().to(torch.float32).contiguous()
    xc, bc, mc, sc, lw, st, tg = f(x), f(basis), f(mean), f(std), f(log_weights), f(start).reshape(-1), f(target).reshape(-1)
    if bc.dim() != 2 or bc.shape[1] != Kx + 2:
        raise ValueError(f"basis must be [samples, {Kx + 2}]; got {tuple(bc.shape)}")
    if mc.shape != sc.shape or mc.dim() != 2 or mc.shape[1] != d or lw.numel() != mc.shape[0]:
        raise ValueError(f"mixture mean/std must be [components, {d}] with one log-weight each; got {tuple(mc.shape)}, "
                         f"{tuple(sc.shape)}, {tuple(lw.shape)}")
    if st.numel() != d or tg.numel() != d:
        raise ValueError(f"start and target poses must have {d} channels")
    Tt = bc.shape[0]
    cost = torch.empty(N, dtype=torch.float32, device=dev)
    traj = torch.empty((N, Tt, d), dtype=torch.float32, device=dev) if want_traj else None
    grad = torch.empty((N, Kx, d), dtype=torch.float32, device=dev) if want_grad else None
    with torch.cuda.device(dev):
        rc = L.sigsvgd_obstacle_cost(xc.data_ptr(), N, Kx, d, st.data_ptr(), tg.data_ptr(), bc.data_ptr(), Tt,
                                     lw.data_ptr(), mc.data_ptr(), sc.data_ptr(), mc.shape[0], float(w_obstacle),
                                     float(w_length), cost.data_ptr(), traj.data_ptr() if want_traj else None,
                                     grad.data_ptr() if want_grad else None, _stream_ptr(dev))
    _lib.check(rc, "obstacle_cost")
    return cost, traj, grad
