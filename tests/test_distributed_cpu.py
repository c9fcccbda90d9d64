"""World-size-2 (and 4) gloo tests of the particle-sharded SVGD iteration: the sharding algebra
(all-gather, folded / cyclic tile ownership, linear partial velocity, reduce-scatter, shard update) must
reproduce the single-process oracle iteration.  The per-rank compute is an oracle-backed double --
on the GPU box the same class runs with the HIP partial solve (tests/test_gpu_api.py)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, N, T, d, steps, q, rowwise=False, fold=True):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import helpers
        from oracle import sigkernel_oracle as O
        from sigsvgd_amd.distributed import ShardedSigSVGD, shard_rows

        X, score = O.synthetic_inputs(N, T, d)
        r0, r1 = shard_rows(N, rank, world)
        Xs, ss = X[r0:r1].clone(), score[r0:r1].clone()
        sh = ShardedSigSVGD(1.0, 0.05, partial_fn=helpers.gram_sym_partial,
                            phi_fn=lambda K, s, gk: helpers.svgd_phi(K, s, gk),
                            rows_fn=lambda Xs_, Xf, ih: helpers.gram_fwd_bwd(Xs_, Xf, ih), rowwise=rowwise, fold=fold)
        for _ in range(steps):
            Xs = sh.step(Xs, ss)
        K = sh.gather_gram()
        q.put((rank, Xs.numpy(), K.numpy() if rank == 0 else None))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,N,rowwise,fold", [(2, 16, False, True), (4, 16, False, True), (2, 16, True, True),
                                                  (2, 20, False, False)])
def test_sharded_iteration_matches_single_process(world, N, rowwise, fold):
    from oracle import sigkernel_oracle as O

    T, d, steps = 6, 2, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 1000) + world + (7 if rowwise else 0) + (13 if not fold else 0)
    procs = [ctx.Process(target=_worker, args=(r, world, port, N, T, d, steps, q, rowwise, fold)) for r in range(world)]
    for p in procs:
        p.start()
    outs = []
    for _ in range(world):  # (a rank that raised never puts: fail as soon as one has exited non-zero)
        for _ in range(300):
            try:
                outs.append(q.get(timeout=1))
                break
            except Exception:
                assert all(p.exitcode in (None, 0) for p in procs), [p.exitcode for p in procs]
        else:
            raise AssertionError("timeout waiting for the ranks")
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    outs.sort(key=lambda t: t[0])
    X_sharded = np.concatenate([o[1] for o in outs], axis=0)
    K_last = outs[0][2]

    X, score = O.synthetic_inputs(N, T, d)
    Xr = X.numpy().astype(np.float64)
    Kprev = None
    for _ in range(steps):
        Kprev = O.gram(Xr.astype(np.float32), Xr.astype(np.float32), O.RBF, 1.0, 0)
        it = O.svgd_iteration(Xr.astype(np.float32), score.numpy(), h=1.0, n=0, lr=0.05)
        Xr = it["X_new"]
    assert np.abs(X_sharded - Xr).max() / np.abs(Xr).max() < 5e-6
    assert np.abs(K_last - Kprev).max() / np.abs(Kprev).max() < 5e-6  # Gram of the last step's input


def test_partials_sum_to_full():
    import helpers
    from oracle import sigkernel_oracle as O

    X, _ = O.synthetic_inputs(20, 5, 3)
    Kf, gf = O.gram_backward(X.numpy(), X.numpy(), None, O.RBF, 1.0, 0)
    for fold in (False, True):
        for stride in (1, 2, 3):
            parts = [helpers.gram_sym_partial(X, 1.0, off, stride, fold=fold) for off in range(stride)]
            assert np.allclose(sum(p[0].numpy().astype(np.float64) for p in parts), Kf, rtol=1e-6)
            assert np.allclose(sum(p[1].numpy() for p in parts), gf, rtol=1e-9, atol=1e-12)


def test_tile_ownership_partitions_and_balances():
    """every tile has exactly one owner; folded ownership gives every rank the same number of pairs of the upper triangle
    whenever the tile pairs divide evenly (mirror of csrc/sig_common.h TileMap, which the GPU tests check against the
    kernels themselves)"""
    from sigsvgd_amd import ops

    for ntile in (1, 2, 5, 16, 31, 128):
        for world in (1, 2, 3, 4, 8):
            for fold in (False, True):
                owned = [ops.owned_tiles(ntile, r, world, fold) for r in range(world)]
                flat = sorted(t for o in owned for t in o)
                assert flat == list(range(ntile)), (ntile, world, fold)
    nw, N = 8, 1024
    for world in (2, 4, 8):
        items = [sum(N - t * nw for t in ops.owned_tiles(N // nw, r, world, True)) for r in range(world)]
        assert max(items) == min(items)
        cyc = [sum(N - t * nw for t in ops.owned_tiles(N // nw, r, world, False)) for r in range(world)]
        assert max(cyc) > 1.02 * (sum(cyc) / world) or world == 2


def test_shard_rows_validation():
    from sigsvgd_amd.distributed import shard_rows

    assert shard_rows(1024, 3, 8) == (384, 512)
    with pytest.raises(ValueError):
        shard_rows(10, 0, 4)
