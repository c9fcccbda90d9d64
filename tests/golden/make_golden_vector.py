"""Generate tests/golden/ref_vector_kernels.npz from the REFERENCE's own vector kernels (run in the
build container; /root/reference never travels to the GPU box).

    cd /root/repo && PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_vector.py

Captured, all by calling the reference classes verbatim on fixed seeded inputs (fp64):
  GaussianKernel, ScaledGaussianKernel, IMQKernel, ScaledIMQKernel   (src/kernels/_kernels.py:64-299)
with a given bandwidth, with the median heuristic, and (scaled variants) with a metric M; plus the
reference's PathSigKernel wiring (src/kernels/_traj_kernels.py:72-144) executed on top of the oracle's
`signatory`-shaped stand-in, because the real `signatory` is not installed -- that fixture pins the
wiring (basepoint, depth, static kernel on signature features), not the signature arithmetic.
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = "/root/reference"
sys.path.insert(0, REF)
sys.dont_write_bytecode = True

import oracle.torch_oracle as oracle_sigkernel  # noqa: E402
from oracle import vector_oracle as VO  # noqa: E402

signatory = types.ModuleType("signatory")


def _signature(path, depth, basepoint=False):
    return torch.from_numpy(VO.signature(path.detach().numpy(), depth, bool(basepoint))).to(path.dtype)


signatory.signature = _signature
sys.modules["sigkernel"] = oracle_sigkernel
sys.modules["signatory"] = signatory

from src.kernels import (GaussianKernel, IMQKernel, PathSigKernel, ScaledGaussianKernel,  # noqa: E402
                         ScaledIMQKernel)

torch.autograd.set_detect_anomaly(False)
out = {}


def npy(t):
    return t.detach().cpu().numpy() if hasattr(t, "detach") else np.asarray(t)


g = torch.Generator().manual_seed(11)
X = torch.randn(7, 5, generator=g, dtype=torch.float64)
Y = X + 0.3 * torch.randn(7, 5, generator=g, dtype=torch.float64)
Mraw = torch.randn(5, 5, generator=g, dtype=torch.float64)
M = Mraw @ Mraw.T / 5 + 0.5 * torch.eye(5, dtype=torch.float64)  # SPD
Mns = M + 0.2 * torch.randn(5, 5, generator=g, dtype=torch.float64)  # not symmetric
out["X"], out["Y"], out["M"], out["Mns"] = npy(X), npy(Y), npy(M), npy(Mns)

for name, ker in [("gauss", GaussianKernel()), ("imq", IMQKernel())]:
    K, dK = ker(X, Y, h=0.8)
    out[f"{name}_h0.8_K"], out[f"{name}_h0.8_dK"] = npy(K), npy(dK)
    K, dK = ker(X, Y)  # median heuristic
    out[f"{name}_med_K"], out[f"{name}_med_dK"] = npy(K), npy(dK)
    out[f"{name}_h0.8_Konly"] = npy(ker(X, Y, h=0.8, compute_grad=False))
    K, dK = ker(X, X, h=1.1)
    out[f"{name}_xx_K"], out[f"{name}_xx_dK"] = npy(K), npy(dK)

for name, ker in [("sgauss", ScaledGaussianKernel()), ("simq", ScaledIMQKernel())]:
    K, dK = ker(X, Y, h=0.8)
    out[f"{name}_I_h0.8_K"], out[f"{name}_I_h0.8_dK"] = npy(K), npy(dK)
    K, dK = ker(X, Y, M=M.clone(), h=0.8)
    out[f"{name}_M_h0.8_K"], out[f"{name}_M_h0.8_dK"] = npy(K), npy(dK)
    K, dK = ker(X, Y, M=M.clone())
    out[f"{name}_M_med_K"], out[f"{name}_M_med_dK"] = npy(K), npy(dK)
    K, dK = ker(X, Y, M=Mns.clone(), h=1.3)
    out[f"{name}_Mns_h1.3_K"], out[f"{name}_Mns_h1.3_dK"] = npy(K), npy(dK)

# flattening of >2-D inputs ([batch, T, d] particles as the planning scripts pass them)
X3 = torch.randn(6, 4, 3, generator=g, dtype=torch.float64)
out["X3"] = npy(X3)
K, dK = GaussianKernel()(X3, X3, h=1.7)
out["gauss_X3_K"], out["gauss_X3_dK"] = npy(K), npy(dK)

# PathSigKernel wiring on the stand-in signatory (paths [batch, length, channels])
P1 = torch.cumsum(0.3 * torch.randn(6, 8, 2, generator=g, dtype=torch.float64), 1)
P2 = torch.cumsum(0.3 * torch.randn(6, 8, 2, generator=g, dtype=torch.float64), 1)
out["P1"], out["P2"] = npy(P1), npy(P2)
psk = PathSigKernel()
K, dK = psk(P1, P2, depth=3, h=0.9)
out["psk_d3_h0.9_K"], out["psk_d3_h0.9_dK"] = npy(K), npy(dK)
K, dK = psk(P1, P2, depth=2)
out["psk_d2_med_K"], out["psk_d2_med_dK"] = npy(K), npy(dK)
out["psk_d3_Konly"] = npy(psk(P1, P1, depth=3, compute_grad=False))

dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_vector_kernels.npz")
np.savez_compressed(dst, **out)
print(f"wrote {dst}: {len(out)} arrays, {os.path.getsize(dst)} bytes")
