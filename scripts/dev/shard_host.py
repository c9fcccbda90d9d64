"""Is the sharded step host-bound at 8 ranks?  One process, nccl world size 1, but the partial solve owns 1/8 of the tiles
(rank 3 of 8): wall-clock per step over 200 steps (one sync at the end) against the GPU time of the same steps (events)."""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, ".")
from sigsvgd_amd import ops
from sigsvgd_amd.distributed import ShardedSigSVGD
from sigsvgd_amd.utils.synthetic import synthetic_inputs

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29544")
dev = torch.device("cuda:0")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
X, s = synthetic_inputs(1024, 64, 7)
Xg, sg = X.to(dev), s.to(dev)
for G in (1, 8):
    part = lambda Xf, inv_h, off, stride, out=None, fold=True: ops.gram_sym_partial(Xf, inv_h, 3 % G, G, out=out, fold=fold)
    sh = ShardedSigSVGD(1.0, 1e-3, partial_fn=part)
    for _ in range(10):
        sh.step(Xg, sg)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(200):
        sh.step(Xg, sg)
    e1.record()
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_wall = time.perf_counter() - t0
    print(f"share 1/{G}: host issue time {t_issue / 200 * 1e3:.3f} ms/step, wall {t_wall / 200 * 1e3:.3f} ms/step, GPU (events) {e0.elapsed_time(e1) / 200:.3f} ms/step")
dist.destroy_process_group()
