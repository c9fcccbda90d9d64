"""Stein variational gradient descent driver with the reference's interface (src/inference/svgd.py:11-159):
`SVGD(kernel, log_p, log_prior, bw_scale, optimizer_class, adaptive_gradient, **opt_args)`,
`.step(X, grad_log_p, optimizer, **kw) -> (X, iter_dict)` and
`.optimize(particles, score_estimator, opt_state, n_steps, debug, callback_func, **kw) -> (data_dict, opt_state)`.

What runs where: the dense part of the velocity, v = -((K @ score - grad_k)/N), is one HIP launch
(`ops.svgd_phi`, fp32 MFMA); with `optimizer_class=None` the same launch also applies the reference's simple
Adagrad (`adaptive_gradient=True`) and the update X - lr*g; with the reference's default optimizer, a plain
`torch.optim.Adam`, the same launch applies Adam's update (`ops.svgd_adam`: exp_avg / exp_avg_sq and the step
counter live in an `ops.AdamState` that is mirrored into the torch optimizer's state, so `optimizer.state_dict()`
and `load_state_dict` keep working, svgd.py:130-133,158).  Other torch optimizers, or Adam with amsgrad /
weight decay / maximize, are driven through the usual closure.  Deliberate differences, none of which changes a result:

  * `iter_dict_device`: "cpu" (default; the reference moves every tensor of the per-iteration dict to the host,
    svgd.py:85-90) or None to leave them on the GPU -- the eager copy of the N x N Gram matrix otherwise
    dominates a millisecond-scale iteration;
  * the particle trace is written into one preallocated [n_steps+1, ...] device buffer and copied to the host
    once, not re-concatenated on the host every iteration (svgd.py:150-152);
  * with `optimizer_class=None` X stays detached between iterations (the reference re-binds it to a non-leaf
    whose autograd graph grows, svgd.py:115,146).
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch
import torch.optim as optim

from .. import ops

_NEEDS_TARGET = (
    "SVGD needs a function to evaluate the log probability of the target distribution or an estimate of the "
    "gradient for every particle."
)


class SVGD:
    """Stein variational gradient descent with pluggable kernels."""

    def __init__(self, kernel=None, log_p: Callable = None, log_prior: Callable = None, bw_scale: float = 1.0,
                 optimizer_class: optim.Optimizer = optim.Adam, adaptive_gradient: bool = False,
                 iter_dict_device="cpu", **opt_args):
        if kernel is None:  # the reference's default (svgd.py:24-25)
            from ..kernels import GaussianKernel

            kernel = GaussianKernel()
        self.kernel, self.log_p, self.log_prior = kernel, log_p, log_prior
        self.bw_scale = bw_scale
        self.optimizer_class, self.opt_args = optimizer_class, opt_args
        self.opt_adagrad, self.opt_inertia = adaptive_gradient, 0
        self.iter_dict_device = iter_dict_device
        self._fuse_manual_update = None

    # ---------------------------------------------------------------------------------------------------
    # kernel term
    # ---------------------------------------------------------------------------------------------------
    def _compute_kernel(self, X: torch.Tensor, **kwargs):
        """(K, grad_k [N, D]): grad_k[i] = sum_j d k(x_i, x_j) / d x_i (derivative in the first slot)."""
        kernel = self.kernel
        if hasattr(kernel, "gram_and_grad"):  # signature kernels: K and its gradient from one fused launch
            k_xx, grad_k = kernel.gram_and_grad(X.detach())
            return k_xx, grad_k.flatten(1)
        if getattr(kernel, "analytic_grad", False):
            k_xx, grad_k = kernel(X, X)
        else:
            Xg = X.detach().requires_grad_(True)
            k_xx = kernel(Xg, Xg.detach(), compute_grad=False)
            grad_k = torch.autograd.grad(k_xx.sum(), Xg)[0].flatten(1)
        return k_xx.detach(), grad_k.detach()

    def _kernel_terms(self, X, kwargs):
        if "k_xx" in kwargs and "grad_k" in kwargs:  # handed in by a score estimator
            k_xx, grad_k = kwargs["k_xx"], kwargs["grad_k"]
            return k_xx, (grad_k.flatten(1) if grad_k.dim() > 1 else grad_k)
        return self._compute_kernel(X, **kwargs)

    # ---------------------------------------------------------------------------------------------------
    # score term
    # ---------------------------------------------------------------------------------------------------
    def _score_terms(self, X, grad_log_p, kwargs):
        """(score [N, D], loss): grad log p of the target (given, or differentiated from log_p) plus the
        gradient of the log prior if there is one."""
        if grad_log_p is None:
            Xg = X.detach().requires_grad_(True)
            log_lik = self.log_p(Xg).sum()
            score = torch.autograd.grad(log_lik, Xg)[0].flatten(1)
            loss = -log_lik.detach()
        else:
            score = grad_log_p.flatten(1)
            loss = kwargs["loss"].sum() if "loss" in kwargs else grad_log_p.norm()
        if self.log_prior is not None:
            Xg = X.detach().requires_grad_(True)
            prior_grad = torch.autograd.grad(self.log_prior(Xg).sum(), Xg)[0]
            score = score + prior_grad.detach().flatten(1)
        return score, loss

    def _export(self, iter_dict: dict) -> dict:
        dev = self.iter_dict_device
        move = (lambda t: t.detach()) if dev is None else (lambda t: t.detach().to(dev))
        return {k: (move(v) if hasattr(v, "detach") else v) for k, v in iter_dict.items()}

    # ---------------------------------------------------------------------------------------------------
    # velocity
    # ---------------------------------------------------------------------------------------------------
    def _velocity(self, X: torch.Tensor, grad_log_p: torch.Tensor, **kwargs) -> Tuple[torch.Tensor, dict]:
        if self.log_p is None and grad_log_p is None:
            raise ValueError(_NEEDS_TARGET)
        k_xx, grad_k = self._kernel_terms(X, kwargs)
        score, loss = self._score_terms(X, grad_log_p, kwargs)

        fuse = self._fuse_manual_update
        if fuse is None:
            velocity = ops.svgd_phi(k_xx, score, grad_k)
        else:  # step(optimizer=None): [Adagrad scaling,] X - lr*g in the same launch (reference svgd.py:108-115)
            velocity, X_new = ops.svgd_phi(k_xx, score, grad_k, X=X.detach(), lr=fuse["lr"],
                                           adagrad_state=fuse["adagrad_state"])
            fuse["X_new"] = X_new.reshape(X.shape).to(X.dtype)
        velocity = velocity.reshape(X.shape).to(X.dtype)

        iter_dict = {"k_xx": k_xx, "grad_k": grad_k, "loss": loss, **kwargs}
        return velocity, self._export(iter_dict)

    # ---------------------------------------------------------------------------------------------------
    # one update
    # ---------------------------------------------------------------------------------------------------
    def _grad_entry(self, g):
        g = g.detach()
        return g if self.iter_dict_device is None else g.to(self.iter_dict_device)

    def _fused_manual_step(self, X, grad_log_p, kwargs):
        """velocity, optional simple Adagrad (state in self.opt_inertia, in place) and X - lr*g: one launch."""
        state = None
        if self.opt_adagrad:
            if not torch.is_tensor(self.opt_inertia):
                self.opt_inertia = torch.zeros(X.shape, dtype=torch.float32, device=X.device)
            state = self.opt_inertia
        self._fuse_manual_update = {"lr": float(self.opt_args["lr"]), "adagrad_state": state}
        try:
            grad, iter_dict = self._velocity(X, grad_log_p, **kwargs)
            X_new = self._fuse_manual_update["X_new"]
        finally:
            self._fuse_manual_update = None
        return X_new, grad, iter_dict

    # ---- torch.optim.Adam fused into the velocity launch ---------------------------------------------------
    def _adam_fusable(self, X, optimizer) -> bool:
        if type(optimizer) is not torch.optim.Adam or type(self)._velocity is not SVGD._velocity:
            return False
        if X.device.type != "cuda" or X.dtype != torch.float32 or not X.is_contiguous() or X.dim() < 2:
            return False
        if len(optimizer.param_groups) != 1 or len(optimizer.param_groups[0]["params"]) != 1:
            return False
        g = optimizer.param_groups[0]
        if g["params"][0] is not X:
            return False
        return not (g.get("amsgrad") or g.get("weight_decay") or g.get("maximize") or g.get("capturable")
                    or g.get("differentiable") or torch.is_tensor(g["lr"]))

    def _adam_state(self, X, optimizer):
        """class-owned state of the fused update, created from (and mirrored back into) the torch optimizer"""
        cached = getattr(self, "_adam_cache", None)
        g = optimizer.param_groups[0]
        if cached is None or cached[0] is not optimizer:
            st = ops.AdamState(X, betas=g["betas"], eps=g["eps"])
            have = optimizer.state.get(X)
            if have:  # resumed from a state_dict (svgd.py:130-133)
                st.exp_avg.copy_(have["exp_avg"].reshape(st.exp_avg.shape))
                st.exp_avg_sq.copy_(have["exp_avg_sq"].reshape(st.exp_avg_sq.shape))
                st.t_host = int(have["step"])
                st.step.fill_(st.t_host)
            self._adam_cache = cached = (optimizer, st)
        return cached[1]

    def _fused_adam_step(self, X, grad_log_p, optimizer, kwargs):
        if self.log_p is None and grad_log_p is None:
            raise ValueError(_NEEDS_TARGET)
        k_xx, grad_k = self._kernel_terms(X, kwargs)
        score, loss = self._score_terms(X, grad_log_p, kwargs)
        st = self._adam_state(X, optimizer)
        with torch.no_grad():
            v, _ = ops.svgd_adam(k_xx, score, grad_k, X, optimizer.param_groups[0]["lr"], st, inplace=True)
        X.grad = v.reshape(X.shape)
        optimizer.state[X] = {"step": torch.tensor(float(st.t_host)), "exp_avg": st.exp_avg.view(X.shape),
                              "exp_avg_sq": st.exp_avg_sq.view(X.shape)}
        info = self._export({"k_xx": k_xx, "grad_k": grad_k, "loss": loss, **kwargs})
        info["grad"] = self._grad_entry(X.grad)
        return X, info

    def step(self, X: torch.Tensor, grad_log_p: torch.Tensor = None, optimizer: optim.Optimizer = None, **kwargs):
        if isinstance(optimizer, torch.optim.Optimizer) and self._adam_fusable(X, optimizer):
            return self._fused_adam_step(X, grad_log_p, optimizer, kwargs)
        if isinstance(optimizer, torch.optim.Optimizer):
            def closure():  # torch optimizers: the velocity is the "gradient" they descend along
                optimizer.zero_grad()
                X.grad, info = self._velocity(X, grad_log_p, **kwargs)
                info["grad"] = self._grad_entry(X.grad)
                return info

            return X, optimizer.step(closure)

        fusable = type(self)._velocity is SVGD._velocity and X.dtype == torch.float32 and X.device.type == "cuda"
        if fusable:
            X, grad, iter_dict = self._fused_manual_step(X, grad_log_p, kwargs)
        else:  # subclasses that post-process the velocity (TrajectorySVGD's mask), other dtypes
            grad, iter_dict = self._velocity(X, grad_log_p, **kwargs)
            if self.opt_adagrad:  # running sum of squared gradients (reference svgd.py:110-113)
                self.opt_inertia = self.opt_inertia + grad**2
                grad = grad / torch.sqrt(self.opt_inertia + 1e-12)
            X = X.detach() - self.opt_args["lr"] * grad
        iter_dict["grad"] = self._grad_entry(grad)
        return X, iter_dict

    # ---------------------------------------------------------------------------------------------------
    # loop
    # ---------------------------------------------------------------------------------------------------
    def _make_optimizer(self, X, opt_state: Optional[dict]):
        if self.optimizer_class is None:
            return None
        optimizer = self.optimizer_class(params=[X], **self.opt_args)
        if opt_state is not None:
            optimizer.load_state_dict(opt_state)
        return optimizer

    def optimize(self, particles: torch.Tensor, score_estimator: Callable = None, opt_state: dict = None,
                 n_steps: int = 100, debug: bool = False, callback_func=None, **kwargs) -> tuple:
        X = particles.detach()
        optimizer = self._make_optimizer(X, opt_state)
        if debug:
            from tqdm import trange

            steps = trange(n_steps, position=0, leave=True)
        else:
            steps = range(n_steps)

        trace = torch.empty((n_steps + 1, *X.shape), dtype=X.dtype, device=X.device)
        trace[0] = X
        data_dict, grad_log_p = {}, None
        for i in steps:
            if score_estimator is not None:
                X.requires_grad_(True)
                grad_log_p, score_dict = score_estimator(X)
                kwargs.update(score_dict)
            X, data_dict[i] = self.step(X, grad_log_p, optimizer, **kwargs)
            trace[i + 1] = X.detach()
            if debug:
                steps.set_postfix(loss=data_dict[i]["loss"].norm(), refresh=False)
            if callback_func is not None:
                callback_func(X)
        data_dict["trace"] = trace.cpu()
        particles[:] = X.detach()  # the reference hands the result back through its input, in place
        return data_dict, (optimizer.state_dict() if optimizer is not None else None)
