"""TEST ONLY: bench.py's process plumbing (self-launched ranks, one JSON line from rank 0, the per-rank phase report,
exit codes) rehearsed on CPU tensors over gloo with the oracle-backed doubles of tests/helpers.py at a toy size.
Run as a script with bench.py's arguments by tests/test_bench_launcher.py; the line it prints is marked "rehearsal"
and is not a measurement.  bench.py itself contains no CPU or oracle-backed compute path."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

import bench  # noqa: E402


class RehearsalBackend:
    label = ("CPU tensors over gloo with oracle-backed test doubles at a toy size: exercises the launcher and the "
             "sharded loop only, NOT a measurement")
    shape = (16, 6, 2)

    def __init__(self, local_rank: int):
        import torch

        import helpers

        self.compute = helpers
        self.dev = torch.device("cpu")
        self.sync = lambda: None

    def init_process_group(self, dist, rank, world):
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(bench._free_port()))
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)

    def sharded(self):
        from sigsvgd_amd.distributed import ShardedSigSVGD

        c = self.compute
        return ShardedSigSVGD(1.0 / bench.H, bench.LR, partial_fn=c.gram_sym_partial, phi_fn=c.svgd_phi,
                              rows_fn=c.gram_fwd_bwd)


if __name__ == "__main__":
    sys.exit(bench.main(backend_cls=RehearsalBackend, script=os.path.abspath(__file__)))
