#!/usr/bin/env python
"""Headline benchmark: SVGD iterations/sec with the signature kernel at N=1024, T=64, d=7
(BASELINE.json metric; config C4 of SURVEY.md §8) on 1..8 MI355X.

One step = one pass of the hot path over the (device-resident) particle batch:
    K = Gram(X, X), grad_k = d sum(K)/dX           HIP: gram_fast_kernel (+ memset, finalize)
    X <- X - lr * v,  v = -((K @ score - grad_k)/N)  HIP: svgd_phi_kernel (fp32 MFMA + fused update)
K and grad_k are materialised in HBM every step (they are API outputs of the reference's
`SVGD.step`); nothing is copied to the host inside the timed region.

N > 1: one process per GPU (torch.distributed, backend nccl = RCCL), particles sharded by rows,
all-gather of X/score + reduce-scatter of v (sigsvgd_amd/distributed.py).  The problem size is
fixed, so scaling is "strong".

Prints ONE JSON line on rank 0 (contract in the task description) with `roofline` (dominant kernel,
HBM bound as BASELINE.json asks, plus the fp64-VALU figure that actually bounds it) and
`cpu_baseline` (the C/OpenMP oracle timed on this box's host cores on a bounded row sample).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

N, T, D_CH, H, LR = 1024, 64, 7, 1.0, 1e-3
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_VALU_PEAK_TF = 78.6       # vector fp64: 256 CU x 4 SIMD x 16 FMA lanes x 2 x 2.4 GHz


def algorithmic_bytes(n, t, d):
    """write K + read K for the GEMM + read X, read score, write grad_k, write phi,
    write X'  =  4*(2 N^2 + 5 N T d) bytes per iteration (17,563,648 B at C4)."""
    return 4 * (2 * n * n + 5 * n * t * d)


def algorithmic_flops(n, t, d, symmetric=True):
    """SURVEY.md §8(d) per ordered pair: T^2(2d+12) + 3P^2 + 18P^2 + 6P^2 + T^2(3d+4); unordered
    pairs (N(N+1)/2) when symmetric, plus the N x N x (T d) GEMM."""
    P = t - 1
    per_pair = t * t * (2 * d + 12) + 27 * P * P + t * t * (3 * d + 4)
    pairs = n * (n + 1) // 2 if symmetric else n * n
    return pairs * per_pair + 2 * n * n * t * d


def cpu_baseline(n, t, d, budget_rows=256):
    """C/OpenMP restatement (oracle/sigkernel_c.c) on all host cores, bounded sample: the first
    `budget_rows` rows of the N x N Gram + gradient (ordered pairs, as the reference computes them),
    scaled by N/rows, plus the dense update."""
    from oracle import c_oracle as C
    from oracle import sigkernel_oracle as O

    X, score = O.synthetic_inputs(n, t, d)
    Xn = X.numpy()
    C.build()
    C.gram_fwd_bwd(Xn, Xn, H, 0, rows=(0, 8))  # warm-up (thread pool, page faults)
    t0 = time.perf_counter()
    K, g = C.gram_fwd_bwd(Xn, Xn, H, 0, rows=(0, budget_rows))
    dt_rows = time.perf_counter() - t0
    import numpy as np

    Kfull = np.zeros((n, n))
    Kfull[:budget_rows] = K
    gfull = np.zeros((n, t, d))
    gfull[:budget_rows] = g
    t0 = time.perf_counter()
    C.svgd_update(Kfull, score.numpy(), gfull, Xn, LR)
    dt_upd = time.perf_counter() - t0
    t_iter = dt_rows * (n / budget_rows) + dt_upd
    return {
        "value": 1.0 / t_iter,
        "unit": "iters/sec",
        "cores": C.num_threads(),
        "kind": "port",
        "sample": f"rows 0..{budget_rows - 1} of {n} (all {n} partners each, ordered pairs, fwd+bwd fp64) "
                  f"in {dt_rows:.2f} s, scaled x{n // budget_rows}, + full dense update {dt_upd * 1e3:.1f} ms; "
                  "anomaly mode off",
        "host_cpus": os.cpu_count(),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-rows", type=int, default=256)
    ap.add_argument("--headline-only", action="store_true",
                    help="skip the other configurations and the eager-copy figure (clean per-kernel profiles)")
    ap.add_argument("--force-dist", action="store_true", help="use the sharded path even with one rank (rehearsal)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    from oracle import sigkernel_oracle as O
    from sigsvgd_amd import _lib, ops

    _lib.load()  # fail loudly if the HIP extension is missing
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    X0, score0 = O.synthetic_inputs(N, T, D_CH)

    use_dist = world > 1 or args.force_dist
    if use_dist:
        import torch.distributed as dist

        from sigsvgd_amd.distributed import ShardedSigSVGD, shard_rows

        dist.init_process_group(backend="nccl", device_id=dev)
        r0, r1 = shard_rows(N, rank, world)
        X = X0[r0:r1].to(dev).contiguous()
        score = score0[r0:r1].to(dev).contiguous()
        sharded = ShardedSigSVGD(1.0 / H, LR)

        def step(Xc):
            return sharded.step(Xc, score)

        def barrier():
            dist.barrier()
            torch.cuda.synchronize()
    else:
        X = X0.to(dev)
        score = score0.to(dev)

        def step(Xc):
            K, gk = ops.gram_fwd_bwd(Xc, Xc, 1.0 / H, 0, y_is_x=True)
            _, Xn = ops.svgd_phi(K, score, gk, X=Xc, lr=LR)
            return Xn

        def barrier():
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        X = step(X)
    barrier()

    # ---- timed region: exactly `steps` iterations, inputs resident in HBM ------------------------
    t0 = time.perf_counter()
    for _ in range(args.steps):
        X = step(X)
    barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        import torch.distributed as dist

        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    assert torch.isfinite(X).all()

    # ---- dominant kernel timed live with HIP events on the launch stream (rank 0, N=1 workload) ----
    roofline = None
    if rank == 0:
        Xe = X0.to(dev)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
        for _ in range(3):
            ops.gram_fwd_bwd(Xe, Xe, 1.0 / H, 0, y_is_x=True)
        torch.cuda.synchronize()
        for a, b in ev:
            a.record()  # torch's current stream == the stream the library launches on
            ops.gram_fwd_bwd(Xe, Xe, 1.0 / H, 0, y_is_x=True)
            b.record()
        torch.cuda.synchronize()
        ms = sorted(a.elapsed_time(b) for a, b in ev)
        k_ms = sum(ms) / len(ms)
        by = algorithmic_bytes(N, T, D_CH)
        fl = algorithmic_flops(N, T, D_CH)
        achieved = by / (k_ms * 1e-3) / 1e9
        roofline = {
            "bound": "hbm",
            "kernel": "gram_fast_kernel<8,8,grad,sym,d=7> (+3.7 MB memset, finalize cast)",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            # HBM-side bytes per launch from rocprofv3 PMC passes (profiles/r01_pmc_hbm_traffic.csv):
            # FETCH_SIZE 162 MB + WRITE_SIZE 1260 MB.  Both are the ~29 M fp64 atomics of the column-side
            # gradient reduction (448 per row tile and column, executed at the memory side and counted
            # ~43 B written / ~5 B fetched each), not re-reads of inputs; they cost ~1.3 % of kernel time
            "traffic": 162e6 + 1.26e9,
            "algorithmic_bytes_per_launch": by,
            "avg_launch_ms": k_ms,
            "median_launch_ms": ms[len(ms) // 2],
            "note": "fused pair solves are fp64-VALU/latency bound, not HBM bound (SURVEY.md §8d); "
                    "secondary figure below",
            # SQ counters of the same kernel (own rocprofv3 --pmc pass, profiles/r01_sq_counters_gram_fast_final.csv):
            # what actually bounds it is vector-instruction issue
            "valu_issue": {"vector_pipe_busy_frac": 0.84, "vector_insts_per_launch": 3.45e9,
                           "source": "profiles/r01_sq_counters_gram_fast_final.csv"},
            "valu_fp64": {
                "achieved_tflops": fl / (k_ms * 1e-3) / 1e12,
                "peak_tflops": FP64_VALU_PEAK_TF,
                "frac": fl / (k_ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TF,
                "algorithmic_flops_per_launch": fl,
            },
        }

    # ---- second figure: the reference's eager per-iteration `.cpu()` of K, grad_k and v
    # (svgd.py:85-90), PCIe-inclusive.  Reported beside `value`, never as `value`.
    eager = None
    if rank == 0 and not use_dist and not args.headline_only:
        Xe2 = X0.to(dev)
        n_e = 20
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n_e):
            K, gk = ops.gram_fwd_bwd(Xe2, Xe2, 1.0 / H, 0, y_is_x=True)
            v, Xe2 = ops.svgd_phi(K, score, gk, X=Xe2, lr=LR)
            _ = (K.cpu(), gk.cpu(), v.cpu())
        torch.cuda.synchronize()
        eager = n_e / (time.perf_counter() - t0)

    # ---- the other BASELINE.json configurations that fit one GPU, a few iterations each --------
    others = None
    if rank == 0 and not use_dist and not args.headline_only:
        others = {}
        for name, (n_, t_, d_, dy) in {"C1": (16, 20, 2, 2), "C2": (128, 32, 7, 0), "C3": (512, 64, 3, 0),
                                       "C5 path shape, N=256 of 4096": (256, 128, 14, 0)}.items():
            Xo, so = O.synthetic_inputs(n_, t_, d_)
            Xo, so = Xo.to(dev), so.to(dev)

            def it(Xc):
                K, gk = ops.gram_fwd_bwd(Xc, Xc, 1.0 / H, dy, y_is_x=True, check_regime=False)
                return ops.svgd_phi(K, so, gk, X=Xc, lr=LR)[1]

            for _ in range(3):
                Xo = it(Xo)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                Xo = it(Xo)
            torch.cuda.synchronize()
            others[name] = {"N": n_, "T": t_, "d": d_, "dyadic_order": dy,
                            "ms_per_iter": (time.perf_counter() - t0) / 10 * 1e3}

    if rank == 0:
        out = {
            "metric": "SVGD iters/sec, sig-kernel N=1024 T=64 d=7",
            "value": args.steps / elapsed,
            "unit": "iters/sec",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "C4: 7-DoF Panda-shaped particles, N=1024 trajectories x T=64 points x d=7, "
                            "sig-kernel PDE dyadic order 0, RBF h=1, fwd+grad+phi+update, lr=1e-3",
                "N": N, "T": T, "d": D_CH,
                "io_dtype": "f32",
                "parallelism": f"particle-sharded x{world}" if world > 1 else "single GPU",
                "host_copies": "none in the timed region (reference-style eager .cpu() of K is opt-in)",
            },
            "eager_cpu_copies_iters_per_sec": eager,
            "other_configs": others,
            "roofline": roofline,
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(N, T, D_CH, args.cpu_rows)
        print(json.dumps(out), flush=True)

    if use_dist:
        import torch.distributed as dist

        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
