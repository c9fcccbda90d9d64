from ._kernels import BaseKernel
from ._traj_kernels import BatchGaussianKernel, SignatureKernel

__all__ = ["BaseKernel", "BatchGaussianKernel", "SignatureKernel"]
