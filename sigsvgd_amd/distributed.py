"""Particle-sharded SVGD iteration over the GPUs of one node (one process per GPU,
`torch.distributed`, backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The reference has no distributed code; this is new design for the path's
natural sharding: N^2/2 independent pair solves followed by one reduction over partners.

    rank r owns particle rows [r*N/G, (r+1)*N/G)
    1. all-gather   X and score shards (2*N*T*d*4 B; 3.7 MB at N=1024,T=64,d=7) into two contiguous operands, issued as
                    ONE grouped RCCL operation (ncclGroupStart / ncclGroupEnd through torch's coalescing manager: a single
                    launch on this latency-bound path); gloo, which has no grouped form, issues them one after the other
    2. compute      the unordered pairs {i <= j} whose row tile (ops.sym_tile_rows(T, d) rows: 4 for T <= 64 with
                    d > 8, else 8) has index r, r + G, ... or is the mirror image ntile-1-t of such a tile
                    (FOLDED ownership: in the upper triangle tile t holds N - t*rows columns, so a tile and
                    its mirror image always hold the same number of pairs and every rank gets the same
                    share; cyclic ownership alone gives the first rank 5.4 % more than the mean at N=1024,
                    G=8), on the gathered X:  K_partial [N,N], grad_partial [N,T,d]
                    v_partial = -((K_partial @ score - grad_partial)/N)   (linear in the partials)
    3. reduce-scatter(sum) v_partial -> this rank's rows of v;  X_shard <- X_shard - lr * v  (one launch)
All per-step buffers (gathered operands, partials, velocity) are allocated once and reused.
K itself stays distributed (each rank keeps its partial; `gather_gram` sums it on demand).

Both collectives are latency-bound at these sizes (a 229 KB shard per peer over a dedicated xGMI
link), so nothing is bucketed or pipelined; the pair solve dominates.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist

from . import ops


def _world(group=None) -> Tuple[int, int]:
    return dist.get_rank(group), dist.get_world_size(group)


def shard_rows(N: int, rank: int, world: int) -> Tuple[int, int]:
    if N % world != 0:
        raise ValueError(f"particle count {N} must be divisible by the number of ranks {world}")
    per = N // world
    return rank * per, (rank + 1) * per


def all_gather_rows(shard: torch.Tensor, group=None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[N/G, ...] -> [N, ...] (rank order)."""
    world = dist.get_world_size(group)
    if out is None:
        out = torch.empty((shard.shape[0] * world,) + tuple(shard.shape[1:]), dtype=shard.dtype, device=shard.device)
    dist.all_gather_into_tensor(out, shard.contiguous(), group=group)
    return out


def all_gather_rows_pair(a: torch.Tensor, b: torch.Tensor, out_a: torch.Tensor, out_b: torch.Tensor, group=None) -> bool:
    """Two all-gathers (same shard shape) into their own contiguous outputs as one grouped collective where the backend has
    one (RCCL: ncclGroupStart / ncclGroupEnd, a single launch); otherwise back to back.  Returns whether the grouped form ran."""
    a, b = a.contiguous(), b.contiguous()
    if dist.get_backend(group) == "nccl" and hasattr(dist, "_coalescing_manager"):
        try:
            with dist._coalescing_manager(group=group, device=a.device, async_ops=False):
                dist.all_gather_into_tensor(out_a, a, group=group)
                dist.all_gather_into_tensor(out_b, b, group=group)
            return True
        except (RuntimeError, TypeError, NotImplementedError):  # a torch build without the grouped form for this op
            pass
    dist.all_gather_into_tensor(out_a, a, group=group)
    dist.all_gather_into_tensor(out_b, b, group=group)
    return False


def reduce_scatter_rows(full: torch.Tensor, group=None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """sum over ranks of [N, ...] -> this rank's [N/G, ...] rows."""
    world = dist.get_world_size(group)
    if out is None:
        out = torch.empty((full.shape[0] // world,) + tuple(full.shape[1:]), dtype=full.dtype, device=full.device)
    if dist.get_backend(group) == "gloo":  # gloo has no reduce_scatter: all-reduce then slice (tests only)
        tmp = full.contiguous().clone()
        dist.all_reduce(tmp, group=group)
        r = dist.get_rank(group)
        out.copy_(tmp[r * out.shape[0]:(r + 1) * out.shape[0]])
        return out
    dist.reduce_scatter_tensor(out, full.contiguous(), group=group)
    return out


class _PhaseClock:
    """Per-phase timing of one sharded step: HIP events on the current stream (the collectives and the
    library's launches are all enqueued there), host clocks for CPU tensors (gloo rehearsal)."""

    def __init__(self, device: torch.device):
        self.gpu = device.type == "cuda"
        self.names = []
        self.marks = [self._now()]

    def _now(self):
        if self.gpu:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            return ev
        import time

        return time.perf_counter()

    def __call__(self, name: str) -> None:
        self.names.append(name)
        self.marks.append(self._now())

    def result(self) -> dict:
        if self.gpu:
            torch.cuda.synchronize()
            return {n: self.marks[k].elapsed_time(self.marks[k + 1]) for k, n in enumerate(self.names)}
        return {n: (self.marks[k + 1] - self.marks[k]) * 1e3 for k, n in enumerate(self.names)}


class ShardedSigSVGD:
    """One SVGD iteration with particles sharded across ranks.

    partial_fn(X_full, inv_h, tile_offset, tile_stride[, out=, fold=]) -> (K_partial, grad_partial) defaults to the
    HIP library's symmetric partial solve with folded tile ownership; phi_fn(K, score, grad_k) -> v to the MFMA velocity
    kernel.  (The CPU tests substitute oracle-backed callables to exercise the sharding algebra under gloo.)"""

    def __init__(self, inv_h: float, lr: float, group=None, partial_fn: Optional[Callable] = None,
                 phi_fn: Optional[Callable] = None, rows_fn: Optional[Callable] = None, rowwise: bool = False,
                 fold: bool = True):
        self.inv_h = float(inv_h)
        self.lr = float(lr)
        self.group = group
        self.fold = bool(fold)
        self.partial_fn = partial_fn or ops.gram_sym_partial
        # `out=` / `fold=` are passed only to callables that take them (the 4-argument contract of rounds 1-2 still works;
        # such a callable owns cyclic tiles and allocates its results)
        import inspect

        try:
            params = inspect.signature(self.partial_fn).parameters
            self._partial_kwargs = ("out" in params and "fold" in params) or any(
                q.kind is inspect.Parameter.VAR_KEYWORD for q in params.values())
        except (TypeError, ValueError):
            self._partial_kwargs = True
        if not self._partial_kwargs and self.fold and partial_fn is not None:
            self.fold = False  # (cyclic ownership is what a 4-argument callable implements)
        self.phi_fn = phi_fn or (lambda K, s, gk: ops.svgd_phi(K, s, gk))
        self.rows_fn = rows_fn or (lambda Xs, Xf, inv_h: ops.gram_fwd_bwd(Xs, Xf, inv_h))
        self.rowwise = bool(rowwise)
        self.last_K_partial = None  # ALIASES the step's preallocated buffer: the next step() overwrites it (clone to keep it)
        self.last_K_rows = None
        self.phase_ms = None  # filled by step(profile=True): milliseconds per phase on this rank
        self.last_gather_grouped = None  # whether the last step's two all-gathers went out as one grouped collective
        self._buf = {}        # per-step buffers, allocated once per (shape, dtype, device)

    def _buffers(self, X_shard: torch.Tensor, world: int):
        key = (tuple(X_shard.shape), X_shard.dtype, X_shard.device, world)
        b = self._buf.get(key)
        if b is None:
            n, T, d = X_shard.shape
            N, dev, dt = n * world, X_shard.device, X_shard.dtype
            b = {
                "X_full": torch.empty((N, T, d), dtype=dt, device=dev),
                "s_full": torch.empty((N, T, d), dtype=dt, device=dev),
                "K_partial": torch.empty((N, N), dtype=dt, device=dev),
                "grad_partial": torch.empty((N, T, d), dtype=torch.float64, device=dev),
                "v_rows": torch.empty((n, T, d), dtype=dt, device=dev),
            }
            self._buf = {key: b}  # one shape at a time: a new shape releases the old buffers
        return b

    def step(self, X_shard: torch.Tensor, score_shard: torch.Tensor, profile: bool = False) -> torch.Tensor:
        """Returns the updated shard X_shard - lr * v_rows (a new tensor; the step's internal buffers are reused by the
        next call).  profile=True brackets the phases with events on the current stream (device tensors) or host clocks
        (CPU rehearsal) and leaves the per-phase milliseconds in `self.phase_ms`; it synchronises, so never use it in a
        timed loop."""
        rank, world = _world(self.group)
        mark = _PhaseClock(X_shard.device) if profile else None
        buf = self._buffers(X_shard, world)
        # one grouped collective into preallocated, contiguous operands (no stack / split copies)
        X_full, s_full = buf["X_full"], buf["s_full"]
        self.last_gather_grouped = all_gather_rows_pair(X_shard, score_shard.to(X_shard.dtype), X_full, s_full, self.group)
        if mark:
            mark("all_gather")
        if self.rowwise or not self._partial_supported(X_full):
            out = self._step_rowwise(X_shard, X_full, s_full)
            if mark:
                mark("rowwise_solve_and_update")
                self.phase_ms = mark.result()
            return out
        if self._partial_kwargs:  # (a user callable written to the 4-argument contract gets no preallocated outputs)
            Kp, gp = self.partial_fn(X_full, self.inv_h, rank, world, out=(buf["K_partial"], buf["grad_partial"]),
                                     fold=self.fold)
        else:
            Kp, gp = self.partial_fn(X_full, self.inv_h, rank, world)
        if mark:
            mark("partial_solve")
        self.last_K_partial = Kp
        v_part = self.phi_fn(Kp, s_full, gp.to(s_full.dtype))  # -((Kp @ s - gp)/N), linear in (Kp, gp)
        if mark:
            mark("velocity")
        v_rows = reduce_scatter_rows(v_part.reshape(X_full.shape), self.group, out=buf["v_rows"])
        if mark:
            mark("reduce_scatter")
        out = torch.add(X_shard, v_rows, alpha=-self.lr)  # one launch
        if mark:
            mark("update")
            self.phase_ms = mark.result()
        return out

    @staticmethod
    def _partial_supported(X_full) -> bool:
        """shapes the register-resident and quadrant symmetric kernels cover (include/sigsvgd_hip.h)"""
        return 3 <= X_full.shape[1] <= 128 and X_full.shape[2] <= 16

    def _step_rowwise(self, X_shard, X_full, s_full):
        """Fallback for shapes outside the symmetric partial solve (e.g. T > 128): each rank solves
        the ordered pairs (own rows) x (all columns), so its rows of K, grad_k and v are complete
        locally and no reduce-scatter is needed -- at twice the pair solves."""
        K_rows, g_rows = self.rows_fn(X_shard, X_full, self.inv_h)
        self.last_K_partial = None
        self.last_K_rows = K_rows
        n_all = X_full.shape[0]
        v_rows = -((K_rows.to(s_full.dtype) @ s_full.flatten(1) - g_rows.flatten(1).to(s_full.dtype)) / n_all)
        return X_shard - self.lr * v_rows.reshape(X_shard.shape)

    def gather_gram(self) -> torch.Tensor:
        """Full K (sum of the partials), on demand -- the per-iteration path never needs it."""
        if self.last_K_partial is None:  # row-wise step: rows are complete, just gather them
            return all_gather_rows(self.last_K_rows, self.group)
        K = self.last_K_partial.clone()
        dist.all_reduce(K, group=self.group)
        return K
