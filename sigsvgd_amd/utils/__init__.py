from .math import bw_median
from .scheduler import CosineScheduler, FactorScheduler, SquareRootScheduler

__all__ = ["bw_median", "SquareRootScheduler", "FactorScheduler", "CosineScheduler"]
