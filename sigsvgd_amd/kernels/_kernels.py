"""Kernel base class: bandwidth-function handling shared by the kernels of the path
(reference src/kernels/_kernels.py:12-61)."""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Callable, Tuple, Union

import torch

from ..utils.math import bw_median

scalar_function = Callable[[torch.Tensor], float]
kernel_output = Union[torch.Tensor, Tuple[torch.Tensor, torch.Tensor]]


class BaseKernel(ABC, torch.nn.Module):
    def __init__(self, bandwidth_fn: scalar_function = None, analytic_grad: bool = True, **kwargs):
        """bandwidth_fn maps the pairwise squared distances to a scalar bandwidth; None selects the
        median heuristic.  Anything that is neither None nor callable is a ValueError."""
        super().__init__(**kwargs)
        self.analytic_grad = analytic_grad
        if bandwidth_fn is None:
            self.get_bandwidth = bw_median
        elif callable(bandwidth_fn):
            self.get_bandwidth = bandwidth_fn
        else:
            raise ValueError(
                "Kernel bandwidth must be a callable scalar function, got " + f"{bandwidth_fn} instead.",
            )

    @abstractmethod
    def __call__(self, X: torch.Tensor, Y: torch.Tensor, compute_grad=True, **kwargs) -> kernel_output:
        pass
