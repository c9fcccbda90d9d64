from .math import bw_median
from .scheduler import CosineScheduler, FactorScheduler, SquareRootScheduler
from .spline import NaturalCubicSpline, create_spline_trajectory, natural_cubic_spline_coeffs

__all__ = ["bw_median", "SquareRootScheduler", "FactorScheduler", "CosineScheduler",
           "NaturalCubicSpline", "natural_cubic_spline_coeffs", "create_spline_trajectory"]
