"""GPU parity: long-path kernel (n=0, 65 <= T <= 128: static kernel streamed on the fly, forward
solution regenerated in the reverse sweep) vs the fp64 oracle, via the C ABI."""
import numpy as np
import pytest
import torch

from oracle import c_oracle as C
from oracle import sigkernel_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _paths(A, T, d, seed, scale=0.05):
    rng = np.random.default_rng(seed)
    return np.cumsum(scale * rng.standard_normal((A, T, d)), axis=1).astype(np.float32)


def _rel(a, b):
    return float(np.abs(np.asarray(a, np.float64) - b).max() / np.abs(b).max())


@pytest.mark.parametrize("A,B,T,d", [(5, 6, 128, 14), (3, 9, 128, 7), (6, 5, 65, 3), (4, 4, 66, 2),
                                     (7, 3, 100, 7), (2, 5, 127, 16), (9, 2, 97, 1)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_stream_fwd_bwd(gpu, A, B, T, d, dtype):
    from sigsvgd_amd import ops

    X, Y = _paths(A, T, d, 1), _paths(B, T, d, 2)
    h = 1.1
    go = np.random.default_rng(3).standard_normal((A, B)).astype(np.float32)
    Kref, gref = C.gram_fwd_bwd(X, Y, h, 0, grad_out=go.astype(np.float64))
    Xg, Yg, gog = (torch.as_tensor(t, device=gpu).to(dtype) for t in (X, Y, go))
    K1 = ops.gram_fwd(Xg, Yg, 1.0 / h)
    K2, g2 = ops.gram_fwd_bwd(Xg, Yg, 1.0 / h, grad_out=gog)
    torch.cuda.synchronize()
    assert _rel(K1.cpu().numpy(), Kref) < TOL
    assert _rel(K2.cpu().numpy(), Kref) < TOL
    assert _rel(g2.cpu().numpy(), gref) < TOL
    if T * d <= 128 * 14:  # the coverage kernel's compact layout tops out at T=128, d=14 (LDS)
        K3, g3 = ops.gram_fwd_bwd(Xg, Yg, 1.0 / h, grad_out=gog, force_generic=True)
        assert _rel(g2.cpu().numpy(), g3.double().cpu().numpy()) < TOL


def test_stream_self_gram_c5_shape(gpu):
    """C5 path shape (T=128, d=14) on the benchmark's synthetic particles, Y is X (sym weighting too)"""
    from sigsvgd_amd import ops

    X, _ = O.synthetic_inputs(24, 128, 14)
    Xg = X.to(gpu)
    K, g = ops.gram_fwd_bwd(Xg, Xg, 1.0, y_is_x=True)
    Kref, gref = C.gram_fwd_bwd(X.numpy(), X.numpy(), 1.0, 0)
    assert _rel(K.cpu().numpy(), Kref) < TOL and _rel(g.cpu().numpy(), gref) < TOL
    K2, g2 = ops.gram_fwd_bwd(Xg, Xg, 1.0, sym=True, y_is_x=True)
    assert _rel(g2.cpu().numpy(), 2 * gref) < TOL
