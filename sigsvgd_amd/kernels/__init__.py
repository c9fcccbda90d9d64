from ._kernels import BaseKernel, GaussianKernel, IMQKernel, ScaledGaussianKernel, ScaledIMQKernel
from ._traj_kernels import BatchGaussianKernel, PathSigKernel, SignatureKernel, TrajectoryKernel

__all__ = [
    "BaseKernel",
    "GaussianKernel",
    "ScaledGaussianKernel",
    "IMQKernel",
    "ScaledIMQKernel",
    "TrajectoryKernel",
    "PathSigKernel",
    "BatchGaussianKernel",
    "SignatureKernel",
]
