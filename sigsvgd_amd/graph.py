"""One SVGD iteration of the signature-kernel hot path as a captured HIP graph (fixed shapes).

Small problems are launch-bound: an eager iteration enqueues two memsets, the Gram kernel, the gradient
finalisation and the velocity/update kernel from Python (~0.16 ms per iteration at N=16..128, whatever the
kernels take).  `GraphedSigSVGD` captures exactly those launches once (torch.cuda.CUDAGraph = hipGraph on ROCm;
the library launches on the capturing stream, all buffers are static) and replays them with one host call.

    g = GraphedSigSVGD(X0, inv_h=1.0, dyadic_order=0, lr=1e-3, update="manual" | "adagrad" | "adam")
    g.score.copy_(grad_log_p)      # the caller's cost side writes the score of the current particles
    g.step()                       # K, grad_k, v refreshed; g.X advanced in place
    g.X, g.K, g.grad_k, g.v        # static device tensors (valid until the next step)

Semantics per update mode are those of `SVGD.step` (reference src/inference/svgd.py:93-116) with
optimizer=None (manual), adaptive_gradient=True (the reference's simple Adagrad) or torch.optim.Adam.
"""
from __future__ import annotations

import torch

from . import ops


class GraphedSigSVGD:
    def __init__(self, X0: torch.Tensor, inv_h: float, dyadic_order: int = 0, lr: float = 1e-3, update: str = "manual",
                 betas=(0.9, 0.999), eps: float = 1e-8, warmup: int = 2):
        if X0.device.type != "cuda" or X0.dtype != torch.float32 or X0.dim() != 3:
            raise ValueError("GraphedSigSVGD needs float32 particles [N, T, d] on the HIP device")
        if update not in ("manual", "adagrad", "adam"):
            raise ValueError(f"unknown update mode {update!r}")
        self.inv_h, self.dyadic_order, self.lr, self.update = float(inv_h), int(dyadic_order), float(lr), update
        self.X = X0.detach().clone().contiguous()
        self.score = torch.zeros_like(self.X)
        self._adagrad = torch.zeros_like(self.X) if update == "adagrad" else None
        self._adam = ops.AdamState(self.X, betas, eps) if update == "adam" else None
        self.K = self.grad_k = self.v = None
        self.iterations = 0
        # warm-up on a side stream (allocates the workspace, loads the code objects), then capture
        side = torch.cuda.Stream(device=self.X.device)
        side.wait_stream(torch.cuda.current_stream(self.X.device))
        saved = self._snapshot()
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                self._iteration()
        torch.cuda.current_stream(self.X.device).wait_stream(side)
        torch.cuda.synchronize(self.X.device)
        self._restore(saved)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self._iteration()
        self._restore(saved)  # capture does not execute, but keep the contract explicit

    def _snapshot(self):
        st = [self.X.clone()]
        if self._adagrad is not None:
            st.append(self._adagrad.clone())
        if self._adam is not None:
            st += [self._adam.exp_avg.clone(), self._adam.exp_avg_sq.clone(), self._adam.step.clone()]
        return st

    def _restore(self, st):
        self.X.copy_(st[0])
        if self._adagrad is not None:
            self._adagrad.copy_(st[1])
        if self._adam is not None:
            self._adam.exp_avg.copy_(st[1])
            self._adam.exp_avg_sq.copy_(st[2])
            self._adam.step.copy_(st[3])
            self._adam.t_host = 0

    def _iteration(self):
        K, gk = ops.gram_fwd_bwd(self.X, self.X, self.inv_h, self.dyadic_order, y_is_x=True, check_regime=False)
        if self.update == "adam":
            v, _ = ops.svgd_adam(K, self.score, gk, self.X, self.lr, self._adam, inplace=True)
        else:
            v, _ = ops.svgd_phi(K, self.score, gk, X=self.X, lr=self.lr, adagrad_state=self._adagrad, inplace=True)
        self.K, self.grad_k, self.v = K, gk, v

    def step(self) -> torch.Tensor:
        self.graph.replay()
        self.iterations += 1
        return self.X
