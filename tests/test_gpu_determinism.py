"""Bit-reproducibility of the Gram + gradient launches: every reduction over pairs runs in an order fixed by the launch
geometry (per-segment / per-item slabs added up by a reduction kernel; no floating-point atomics), so two launches on the
same input return the same bits -- eager or replayed from a captured graph, alone on the chip or next to other work.
Rounds 1-2 reduced the gradient with fp32 / fp64 atomics; the driver's round-2 GPU run failed on exactly that."""
import pytest
import torch

from sigsvgd_amd.utils.synthetic import synthetic_inputs

pytestmark = pytest.mark.gpu

# register-resident kernel (8- and 4-row tiles, 32-slot ring, d > 8), quadrant kernel (both channel layouts, the global row
# accumulator of d = 15, 16), coverage kernel (ordered pairs, and the symmetric solve from 4096 pairs on)
SHAPES = [((96, 64, 7), 0), ((80, 32, 7), 0), ((40, 48, 12), 0), ((300, 20, 3), 0), ((24, 128, 14), 0), ((20, 100, 7), 0),
          ((12, 70, 16), 0), ((16, 20, 2), 2), ((72, 10, 2), 3),
          # round 4 (ADVICE): the band kernel (129 .. 256 refined cells) and the refined-grid kernel at dyadic order 5 / 6, whose
          # forced flush of the block sums (r lanes of a row block on one coarse cell) became a fixed butterfly
          ((100, 10, 2), 4), ((35, 30, 2), 3), ((30, 5, 2), 5), ((20, 3, 7), 6), ((24, 33, 9), 1)]


def _disturb(dev):
    """unrelated work on another stream, so that the two launches under test see different machine states"""
    side = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(side):
        a = torch.randn(2048, 2048, device=dev)
        for _ in range(4):
            a = a @ a * 1e-3
    return side, a


@pytest.mark.parametrize("shape,dyadic", SHAPES)
def test_same_input_same_bits(gpu, shape, dyadic):
    from sigsvgd_amd import ops

    X, _ = synthetic_inputs(*shape)
    Xg = X.to(gpu)
    go = torch.randn(shape[0], shape[0], device=gpu)
    for sym_solve in (True, False):
        for weights in (None, go):
            K0, g0 = ops.gram_fwd_bwd(Xg, Xg, 1.0, dyadic, grad_out=weights, y_is_x=sym_solve)
            torch.cuda.synchronize()
            side, keep = _disturb(gpu)
            for _ in range(3):
                K1, g1 = ops.gram_fwd_bwd(Xg, Xg.clone(), 1.0, dyadic, grad_out=weights, y_is_x=sym_solve)
                assert torch.equal(K0, K1) and torch.equal(g0, g1), (shape, sym_solve, weights is not None)
            side.synchronize()
            del keep


@pytest.mark.parametrize("shape", [(96, 64, 7), (40, 32, 3), (24, 128, 14)])
def test_partial_solve_same_bits_and_no_stale_workspace(gpu, shape):
    """the sharded partial solve is reproducible too, and its outputs do not depend on what an earlier launch of another
    shape left in the shared workspace (nothing is accumulated across launches)"""
    from sigsvgd_amd import ops

    X, _ = synthetic_inputs(*shape)
    Xg = X.to(gpu)
    ref = [ops.gram_sym_partial(Xg, 1.0, r, 3) for r in range(3)]
    other, _ = synthetic_inputs(50, 40, 5)
    ops.gram_fwd_bwd(other.to(gpu) * 7.0, other.to(gpu) * 7.0, 1.0, 0, y_is_x=True)  # dirties the workspace
    for r in range(3):
        Kp, gp = ops.gram_sym_partial(Xg, 1.0, r, 3)
        assert torch.equal(Kp, ref[r][0]) and torch.equal(gp, ref[r][1])
    # reused output buffers: K_partial is re-zeroed, grad_partial overwritten
    out = (torch.full_like(ref[0][0], 3.0), torch.full_like(ref[0][1], -5.0))
    Kp, gp = ops.gram_sym_partial(Xg, 1.0, 1, 3, out=out)
    assert Kp is out[0] and torch.equal(Kp, ref[1][0]) and torch.equal(gp, ref[1][1])
    # the shares add up to the full symmetric launch: K exactly, the gradient to fp64 rounding
    K, g = ops.gram_fwd_bwd(Xg, Xg, 1.0, 0, y_is_x=True)
    assert torch.equal(sum(r[0] for r in ref), K)
    gs = sum(r[1] for r in ref)
    assert float((gs - g.double()).abs().max() / g.double().abs().max()) < 1e-6


def test_two_captured_graphs_step_only_the_second(gpu):
    """two iterations captured back to back share the library workspace; replaying only the second gives what the eager
    launches give (ABI <= 7 relied on a zeroed workspace here: a graph that was never replayed never cleaned it)"""
    from sigsvgd_amd import ops
    from sigsvgd_amd.graph import GraphedSigSVGD

    Xa, sa = synthetic_inputs(24, 32, 7)
    Xb, sb = synthetic_inputs(20, 64, 3)
    Xa, sa, Xb, sb = Xa.to(gpu), sa.to(gpu), Xb.to(gpu), sb.to(gpu)
    ga = GraphedSigSVGD(Xa, inv_h=1.0, lr=1e-3, update="manual")  # never replayed
    gb = GraphedSigSVGD(Xb, inv_h=1.0, lr=1e-3, update="manual")
    gb.score.copy_(sb)
    gb.step()
    K, gk = ops.gram_fwd_bwd(Xb, Xb, 1.0, 0, y_is_x=True)
    _, Xn = ops.svgd_phi(K, sb, gk, X=Xb, lr=1e-3)
    torch.cuda.synchronize()
    assert torch.equal(gb.K, K) and torch.equal(gb.grad_k, gk) and torch.equal(gb.X, Xn)
    assert ga.iterations == 0


@pytest.mark.parametrize("N,T,d,scale,h", [(40, 64, 1, 0.2, 0.1), (30, 100, 2, 0.3, 0.3), (48, 32, 3, 0.1, 0.1)])
def test_flagged_pairs_and_their_exact_pass_are_reproducible(gpu, N, T, d, scale, h):
    """rough paths in few channels: many pairs are flagged (cancellation / conditioning) and solved again by the coverage
    kernel's fp64 pass after the launch -- the flags, the repaired entries and the gradient are the same bits call after call"""
    import numpy as np

    from sigsvgd_amd import ops

    rng = np.random.default_rng(N + T)
    X = torch.as_tensor(np.cumsum(scale * rng.standard_normal((N, T, d)), axis=1).astype(np.float32), device=gpu)
    outs = []
    for _ in range(3):
        side, keep = _disturb(gpu)
        outs.append(ops.gram_fwd_bwd(X, X, 1.0 / h, 0, y_is_x=True) + (ops.gram_fwd(X, X.clone(), 1.0 / h, 0),))
        side.synchronize()
        del keep
    for K, g, Kf in outs[1:]:
        assert torch.equal(K, outs[0][0]) and torch.equal(g, outs[0][1]) and torch.equal(Kf, outs[0][2])
