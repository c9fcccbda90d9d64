"""Median bandwidth heuristic, used when a kernel is built with `bandwidth_fn=None`
(reference API: src/utils/math.py:28-34)."""
from __future__ import annotations

import torch


def bw_median(sq_dists: torch.Tensor, bw_scale: float = 1.0, tol: float = 1.0e-8) -> torch.Tensor:
    """bw_scale * sqrt(median(sq_dists) / log(rows + 1)), never below `tol`.

    `torch.median` of the flattened tensor is the LOWER median; rows = sq_dists.shape[0].  The divisor is the
    log of a float32 0-dim tensor, as in the reference, so it carries float32 rounding (the fixtures pin that)."""
    divisor = torch.log(torch.tensor(float(sq_dists.shape[0]) + 1.0))
    bandwidth = bw_scale * torch.sqrt(torch.median(sq_dists) / divisor)
    return bandwidth.clamp_min_(tol)
