from .svgd import SVGD
from .trajectory_svgd import TrajectorySVGD
from .score import ScoreEstimator

__all__ = ["SVGD", "TrajectorySVGD", "ScoreEstimator"]
