"""`python bench.py --gpus N` must start its own ranks (the driver's SCALE run may invoke it without a
launcher) and must also run as one rank under `torch.distributed.run`.  Rehearsed here on CPU tensors
over gloo with the oracle-backed doubles (tests/bench_rehearsal.py drives bench.main with a stand-in backend;
bench.py itself has no CPU path): what is checked is the process
plumbing -- child ranks, one JSON line from rank 0, the per-rank phase report, exit codes -- not a number."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "tests", "bench_rehearsal.py")


def _json_line(stdout: str) -> dict:
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def _check(out: dict, world: int):
    assert out["n_gpus"] == world and out["steps"] == 2 and out["warmup"] == 1
    assert out["metric"].startswith("SVGD iters/sec") and out["unit"] == "iters/sec" and out["value"] > 0
    assert out["scaling"] == "strong" and out["higher_is_better"] is True
    assert "rehearsal" in out  # never mistaken for a measurement
    sh = out["sharded"]
    assert sh["ranks"] == world and sh["rows_per_rank"] * world == out["config"]["N"]
    for key in ("all_gather", "partial_solve", "velocity", "reduce_scatter", "update"):
        assert len(sh["per_rank_ms"][key]) == world
    assert sh["all_gather_us_max"] > 0 and sh["reduce_scatter_us_max"] > 0 and sh["partial_solve_ms_max"] > 0


def test_bench_starts_its_own_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    _check(_json_line(p.stdout), 2)


def test_bench_runs_as_a_rank_under_the_launcher():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), BENCH, "--gpus", "2", "--steps", "2",
                        "--warmup", "1"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    _check(_json_line(p.stdout), 2)


def test_bench_reports_a_failing_rank():
    """3 ranks cannot shard the 16 rehearsal particles: every rank raises, the parent must not exit 0"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, BENCH, "--gpus", "3", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode != 0
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]


def test_committed_counters_parse():
    """bench.py takes `roofline.traffic` / `valu_issue` from the committed rocprofv3 summaries, not from literals"""
    sys.path.insert(0, ROOT)
    import bench

    c = bench.committed_counters()
    if os.path.exists(bench.PMC_TRAFFIC_CSV):
        assert c["traffic"] and c["traffic"] > 0 and "gram_fast_kernel:WRITE_SIZE" in c["traffic_detail"]["per_kernel_raw"]
    else:
        assert c["traffic"] is None
    if os.path.exists(bench.PMC_SQ_CSV):
        v = c["valu_issue"]
        assert v["vector_insts_per_launch"] > 1e8 and 0 < v["wave_frac_issuing_valu"] < 1
    else:
        assert c["valu_issue"] is None
