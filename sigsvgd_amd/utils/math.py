"""Bandwidth heuristic used when `bandwidth_fn is None` (reference src/utils/math.py:28-34)."""
from __future__ import annotations

import torch


def bw_median(sq_dists: torch.Tensor, bw_scale: float = 1.0, tol: float = 1.0e-8) -> torch.Tensor:
    """h = bw_scale * sqrt(median(sq_dists) / log(rows + 1)), clamped at `tol`.

    `torch.median` over the flattened tensor (lower median), rows = sq_dists.shape[0]."""
    h = torch.median(sq_dists)
    h = h / torch.tensor(sq_dists.shape[0] + 1.0).log()
    h = bw_scale * h.sqrt()
    return h.clamp_min_(tol)
