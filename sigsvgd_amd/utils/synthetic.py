"""Synthetic particles of the benchmark contract: generated with torch's CPU generator
so that every box sees bit-identical inputs.  `tests/test_host_logic.py` checks that the oracle's own
generator (oracle/sigkernel_oracle.py) produces the same tensors."""
from __future__ import annotations

import torch


def synthetic_inputs(N: int, T: int, d: int, seed_x: int = 0, seed_s: int = 1):
    """X = cumsum(0.05 * randn(N, T, d), dim=1) in fp64 -> fp32;  score = randn(N, T, d) fp32."""
    gx = torch.Generator(device="cpu").manual_seed(seed_x)
    X = torch.cumsum(0.05 * torch.randn(N, T, d, generator=gx, dtype=torch.float64), dim=1).float()
    gs = torch.Generator(device="cpu").manual_seed(seed_s)
    score = torch.randn(N, T, d, generator=gs, dtype=torch.float32)
    return X, score
