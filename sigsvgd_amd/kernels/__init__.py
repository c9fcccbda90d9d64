"""Kernels of the reference's `src.kernels` package, backed by the HIP library."""
from ._kernels import BaseKernel, GaussianKernel, IMQKernel, ScaledGaussianKernel, ScaledIMQKernel
from ._traj_kernels import BatchGaussianKernel, PathSigKernel, SignatureKernel, TrajectoryKernel

__all__ = sorted(
    ["BaseKernel", "BatchGaussianKernel", "GaussianKernel", "IMQKernel", "PathSigKernel", "ScaledGaussianKernel",
     "ScaledIMQKernel", "SignatureKernel", "TrajectoryKernel"]
)
