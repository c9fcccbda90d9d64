#!/usr/bin/env python
"""Headline benchmark: SVGD iterations/sec with the signature kernel at N=1024, T=64, d=7
(BASELINE.json metric; config C4 of SURVEY.md §8) on 1..8 MI355X.

One step = one pass of the hot path over the (device-resident) particle batch:
    K = Gram(X, X), grad_k = d sum(K)/dX           HIP: gram_fast_kernel + grad_reduce_kernel (fixed-order sums)
    X <- X - lr * v,  v = -((K @ score - grad_k)/N)  HIP: svgd_phi_kernel (fp32 MFMA + fused update)
K and grad_k are materialised in HBM every step (they are API outputs of the reference's
`SVGD.step`); nothing is copied to the host inside the timed region.

N > 1: one process per GPU (torch.distributed, backend nccl = RCCL), particles sharded by rows,
all-gather of X/score + reduce-scatter of v (sigsvgd_amd/distributed.py).  The problem size is
fixed, so scaling is "strong".  `python bench.py --gpus N` without a launcher starts the N ranks
itself (a `torch.distributed.run` child process, started before this process touches the GPU) and
forwards rank 0's JSON line; under `torch.distributed.run` (WORLD_SIZE set) it is one of the ranks.

Prints ONE JSON line on rank 0 (contract in the task description) with `roofline` (dominant kernel,
HBM bound as BASELINE.json asks, plus the fp64-VALU figure that actually bounds it), `cpu_baseline`
(the C/OpenMP oracle timed on this box's host cores on a bounded row sample) and, for sharded runs,
`sharded` (per-rank partial-solve / velocity milliseconds, all-gather and reduce-scatter microseconds).
"""
from __future__ import annotations

import argparse
import csv
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N, T, D_CH, H, LR = 1024, 64, 7, 1.0, 1e-3
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_VALU_PEAK_TF = 78.6       # vector fp64: 256 CU x 4 SIMD x 16 FMA lanes x 2 x 2.4 GHz
PROFILES = os.path.join(ROOT, "profiles")
PMC_TRAFFIC_CSV = os.path.join(PROFILES, "r04_pmc_hbm_traffic.csv")   # written by scripts/pmc_summary.py
PMC_SQ_CSV = os.path.join(PROFILES, "r04_sq_counters_gram_fast.csv")


class HipBackend:
    """What the benchmark runs on: the HIP library on the MI355X of this rank, RCCL between ranks.  (The process-plumbing
    tests under tests/ drive `main()` with a CPU/gloo stand-in of this class; nothing in this file knows about it.)"""

    label = None            # a stand-in sets a text here; it is attached to the JSON line as "rehearsal"
    shape = (N, T, D_CH)
    dist_backend = "nccl"

    def __init__(self, local_rank: int):
        import torch

        from sigsvgd_amd import _lib, ops

        _lib.load()  # fail loudly if the HIP extension is missing
        self.compute = ops
        self.dev = torch.device("cuda", local_rank)
        torch.cuda.set_device(self.dev)
        self.sync = torch.cuda.synchronize

    def init_process_group(self, dist, rank, world):
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        dist.init_process_group(backend=self.dist_backend, rank=rank, world_size=world, device_id=self.dev)

    def sharded(self):
        from sigsvgd_amd.distributed import ShardedSigSVGD

        return ShardedSigSVGD(1.0 / H, LR)


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
    scaled by N/rows, plus the dense update.  The only place bench.py touches oracle/."""
    import numpy as np

    from oracle import c_oracle as C
    from sigsvgd_amd.utils.synthetic import synthetic_inputs

    X, score = synthetic_inputs(n, t, d)
    Xn = X.numpy()
    C.build()
    C.gram_fwd_bwd(Xn, Xn, H, 0, rows=(0, 8))  # warm-up (thread pool, page faults)
    t0 = time.perf_counter()
    K, g = C.gram_fwd_bwd(Xn, Xn, H, 0, rows=(0, budget_rows))
    dt_rows = time.perf_counter() - t0
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


def committed_counters():
    """HBM-side traffic and vector-issue counters of the dominant kernel, read from the rocprofv3 --pmc
    summaries committed under profiles/ (own passes of `scripts/one_gram.py`; the CSVs name the kernel
    instantiation and the git revision they were taken on).  None when a file is absent."""
    out = {"traffic": None, "valu_issue": None}
    try:
        # the dominant "kernel" of an iteration is two launches since round 3: the pair kernel writes the gradient's partial
        # sums, grad_reduce_kernel reads them back and adds them in a fixed order -- both are counted
        with open(PMC_TRAFFIC_CSV) as f:
            rows = [r for r in csv.DictReader(f) if "gram_fast_kernel" in r["kernel"] or "grad_reduce_kernel" in r["kernel"]]
        per = {}
        for r in rows:
            name = "gram_fast_kernel" if "gram_fast_kernel" in r["kernel"] else "grad_reduce_kernel"
            per.setdefault((name, r["counter"]), []).append(float(r["bytes_per_dispatch"]))
        mean = {k: sum(v) / len(v) for k, v in per.items()}
        if all((k, c) in mean for k in ("gram_fast_kernel", "grad_reduce_kernel") for c in ("FETCH_SIZE", "WRITE_SIZE")):
            fetch = mean[("gram_fast_kernel", "FETCH_SIZE")] + mean[("grad_reduce_kernel", "FETCH_SIZE")]
            write = mean[("gram_fast_kernel", "WRITE_SIZE")] + mean[("grad_reduce_kernel", "WRITE_SIZE")]
            # gfx950: FETCH_SIZE tallies 128-B read requests at 64 B (MI355X_MICROARCH.md, HBM); calibrated on this path's
            # own access pattern: the reduction kernel reads a known 129.4 MB at C4 and the counter says 64.7 MB, while
            # WRITE_SIZE matches the known 133.6 MB of the pair kernel exactly -> reads are doubled
            out["traffic"] = 2.0 * fetch + write
            out["traffic_detail"] = {"fetch_bytes_raw": fetch, "write_bytes_raw": write, "fetch_correction": 2.0,
                                     "per_kernel_raw": {f"{k}:{c}": v for (k, c), v in sorted(mean.items())},
                                     "revision": rows[0].get("revision"),
                                     "source": os.path.relpath(PMC_TRAFFIC_CSV, ROOT)}
    except (OSError, KeyError, ValueError):
        pass
    try:
        with open(PMC_SQ_CSV) as f:
            rows = [r for r in csv.DictReader(f) if "gram_fast_kernel" in r["Kernel_Name"]]
        if rows:
            avg = lambda k: sum(float(r[k]) for r in rows) / len(rows)
            out["valu_issue"] = {
                "vector_insts_per_launch": avg("SQ_INSTS_VALU"),
                "wave_frac_issuing_valu": avg("SQ_ACTIVE_INST_VALU") / avg("SQ_WAVE_CYCLES"),
                "wave_frac_parked": avg("SQ_WAIT_ANY") / avg("SQ_WAVE_CYCLES"),
                "wave_frac_issue_stalled": avg("SQ_WAIT_INST_ANY") / avg("SQ_WAVE_CYCLES"),
                "kernel": rows[0]["Kernel_Name"], "vgpr_count_field": rows[0].get("VGPR_Count"),
                "source": os.path.relpath(PMC_SQ_CSV, ROOT)}
    except (OSError, KeyError, ValueError, ZeroDivisionError):
        pass
    return out


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-rows", type=int, default=256)
    ap.add_argument("--headline-only", action="store_true",
                    help="skip the other configurations and the eager-copy figure (clean per-kernel profiles)")
    ap.add_argument("--force-dist", action="store_true", help="use the sharded path even with one rank")
    return ap.parse_args(argv)


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args, script=None) -> int:
    """`python bench.py --gpus N` outside a launcher: start the N ranks as ONE child process tree
    (`python -m torch.distributed.run`, rendezvous on 127.0.0.1) and forward their output.  This process has
    not imported torch.cuda nor made any HIP call, so nothing GPU-initialised is ever exec'ed or forked."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), script or os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
    proc = subprocess.run(cmd, env=env)
    return proc.returncode


def main(argv=None, backend_cls=HipBackend, script=None) -> int:
    args = parse_args(argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args, script)
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")

    import torch

    from sigsvgd_amd.utils.synthetic import synthetic_inputs

    be = backend_cls(local_rank)
    rehearse = be.label is not None
    n, t, d = be.shape
    compute, dev, sync = be.compute, be.dev, be.sync
    X0, score0 = synthetic_inputs(n, t, d)

    use_dist = world > 1 or args.force_dist or rehearse
    sharded = None
    if use_dist:
        import torch.distributed as dist

        from sigsvgd_amd.distributed import shard_rows

        be.init_process_group(dist, rank, world)
        sharded = be.sharded()
        r0, r1 = shard_rows(n, rank, world)
        X = X0[r0:r1].to(dev).contiguous()
        score = score0[r0:r1].to(dev).contiguous()

        def step(Xc):
            return sharded.step(Xc, score)

        def barrier():
            dist.barrier()
            sync()
    else:
        X = X0.to(dev)
        score = score0.to(dev)

        def step(Xc):
            K, gk = compute.gram_fwd_bwd(Xc, Xc, 1.0 / H, 0, y_is_x=True)
            _, Xn = compute.svgd_phi(K, score, gk, X=Xc, lr=LR)
            return Xn

        def barrier():
            sync()

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
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    assert torch.isfinite(X).all()

    # ---- sharded runs: where the step's time goes on every rank (outside the timed region) -------------
    shard_report = None
    if use_dist:
        n_prof = 5
        acc = {}
        for _ in range(n_prof):
            X = sharded.step(X, score, profile=True)
            for k, v in sharded.phase_ms.items():
                acc[k] = acc.get(k, 0.0) + v / n_prof
        gathered = [None] * world
        dist.all_gather_object(gathered, acc)
        if rank == 0:
            keys = list(gathered[0].keys())
            shard_report = {
                "ranks": world,
                "rows_per_rank": n // world,
                "per_rank_ms": {k: [round(g[k], 4) for g in gathered] for k in keys},
                "all_gather_us_max": round(max(g["all_gather"] for g in gathered) * 1e3, 1),
                "reduce_scatter_us_max": round(max(g.get("reduce_scatter", 0.0) for g in gathered) * 1e3, 1),
                "partial_solve_ms_max": round(max(g.get("partial_solve", 0.0) for g in gathered), 4),
                "payload_bytes": {"all_gather_out": 2 * n * t * d * 4, "reduce_scatter_in": n * t * d * 4},
                "note": f"mean of {n_prof} instrumented steps after the timed region (events on the launch stream; "
                        "each instrumented step synchronises)",
            }

    # ---- dominant kernel timed live with HIP events on the launch stream (rank 0, N=1 workload) ----
    roofline = None
    if rank == 0 and not rehearse:
        Xe = X0.to(dev)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
        for _ in range(3):
            compute.gram_fwd_bwd(Xe, Xe, 1.0 / H, 0, y_is_x=True)
        torch.cuda.synchronize()
        for a, b in ev:
            a.record()  # torch's current stream == the stream the library launches on
            compute.gram_fwd_bwd(Xe, Xe, 1.0 / H, 0, y_is_x=True)
            b.record()
        torch.cuda.synchronize()
        ms = sorted(a.elapsed_time(b) for a, b in ev)
        k_ms = sum(ms) / len(ms)
        by = algorithmic_bytes(N, T, D_CH)
        fl = algorithmic_flops(N, T, D_CH)
        achieved = by / (k_ms * 1e-3) / 1e9
        pmc = committed_counters()
        roofline = {
            "bound": "hbm",
            "kernel": "sigsvgd::gram_fast_kernel<8, 8, true, true, true> (DPAD=8, 8 waves, gradient, symmetric, "
                      "d=DPAD-1) + grad_reduce_kernel (fixed-order sums of the gradient partials)",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            # HBM-side bytes per launch (pair kernel + reduction kernel): 2 x FETCH_SIZE + WRITE_SIZE of separate
            # rocprofv3 --pmc passes, read from the committed summary (null if it is missing); see committed_counters()
            "traffic": pmc["traffic"],
            "traffic_detail": pmc.get("traffic_detail"),
            "algorithmic_bytes_per_launch": by,
            "avg_launch_ms": k_ms,
            "median_launch_ms": ms[len(ms) // 2],
            "note": "fused pair solves are bound by vector-instruction issue (fp64 static kernel, fp32 sweeps and "
                    "contraction), not by HBM (SURVEY.md §8d); secondary figures below",
            "valu_issue": pmc["valu_issue"],
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
            K, gk = compute.gram_fwd_bwd(Xe2, Xe2, 1.0 / H, 0, y_is_x=True)
            v, Xe2 = compute.svgd_phi(K, score, gk, X=Xe2, lr=LR)
            _ = (K.cpu(), gk.cpu(), v.cpu())
        torch.cuda.synchronize()
        eager = n_e / (time.perf_counter() - t0)

    # ---- the other BASELINE.json configurations that fit one GPU, a few iterations each --------
    others = None
    if rank == 0 and not use_dist and not args.headline_only:
        others = {}
        for name, (n_, t_, d_, dy) in {"C1": (16, 20, 2, 2), "C2": (128, 32, 7, 0), "C3": (512, 64, 3, 0),
                                       "C5 path shape, N=256 of 4096": (256, 128, 14, 0),
                                       # the reference's own experiment shapes, for the record
                                       "reference notebook experiment (script_sequential_distribution.ipynb)": (100, 10, 2, 4),
                                       "reference maze controller (script_control_particle_maze.py)": (35, 30, 2, 3),
                                       "reference planning experiment (script_planning_obstacle_field.py)": (30, 5, 2, 5)}.items():
            Xo, so = synthetic_inputs(n_, t_, d_)
            Xo, so = Xo.to(dev), so.to(dev)

            def it(Xc):
                K, gk = compute.gram_fwd_bwd(Xc, Xc, 1.0 / H, dy, y_is_x=True, check_regime=False)
                return compute.svgd_phi(K, so, gk, X=Xc, lr=LR)[1]

            for _ in range(5):
                Xo = it(Xo)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                Xo = it(Xo)
            torch.cuda.synchronize()
            others[name] = {"N": n_, "T": t_, "d": d_, "dyadic_order": dy,
                            "ms_per_iter": (time.perf_counter() - t0) / 10 * 1e3}
            if t_ <= 64:  # launch-bound sizes: the same iteration replayed from a captured HIP graph
                from sigsvgd_amd.graph import GraphedSigSVGD

                gr = GraphedSigSVGD(Xo, 1.0 / H, dy, LR, "manual")
                gr.score.copy_(so)
                for _ in range(5):
                    gr.step()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(50):
                    gr.step()
                torch.cuda.synchronize()
                others[name]["ms_per_iter_hip_graph"] = (time.perf_counter() - t0) / 50 * 1e3

        # C5 at its full size on ONE GPU (its home is 8 GPUs): one warm-up + two timed Gram + gradient launches
        Xo, _ = synthetic_inputs(4096, 128, 14)
        Xo = Xo.to(dev)
        compute.gram_fwd_bwd(Xo, Xo, 1.0 / H, 0, y_is_x=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2):
            compute.gram_fwd_bwd(Xo, Xo, 1.0 / H, 0, y_is_x=True)
        torch.cuda.synchronize()
        others["C5 full size on one GPU"] = {"N": 4096, "T": 128, "d": 14, "dyadic_order": 0,
                                             "ms_per_gram_and_gradient": (time.perf_counter() - t0) / 2 * 1e3}
        del Xo

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
            # arithmetic types on the path: static kernel and 4-corner increments fp64; both PDE sweeps fp32 in difference
            # form; stored D / K_fwd / G, the gradient contraction and the velocity GEMM fp32; reduction over partners
            # fp64 in a fixed order; fp32 I/O
            "dtype": "f64 static kernel + f32 sweeps and contraction (mixed), f32 I/O",
            "data": "synthetic",
            "config": {
                "workload": "C4: 7-DoF Panda-shaped particles, N=1024 trajectories x T=64 points x d=7, "
                            "sig-kernel PDE dyadic order 0, RBF h=1, fwd+grad+phi+update, lr=1e-3",
                "N": n, "T": t, "d": d,
                "io_dtype": "f32",
                "parallelism": f"particle-sharded x{world}" if use_dist else "single GPU",
                "host_copies": "none in the timed region (reference-style eager .cpu() of K is opt-in)",
            },
            "eager_cpu_copies_iters_per_sec": eager,
            "other_configs": others,
            "roofline": roofline,
            "sharded": shard_report,
        }
        if rehearse:
            out["rehearsal"] = be.label
            out["config"]["workload"] = f"rehearsal N={n} T={t} d={d}"
        if not args.no_cpu_baseline and world == 1 and not rehearse:
            out["cpu_baseline"] = cpu_baseline(N, T, D_CH, args.cpu_rows)
        print(json.dumps(out), flush=True)

    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
