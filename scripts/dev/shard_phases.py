"""Phase times of one sharded step at world size 1 under nccl (what is NOT the partial solve is overhead that stays
constant as ranks are added).  usage (GPU box): python scripts/dev/shard_phases.py"""
import os, sys, time
sys.path.insert(0, '.')
import torch, torch.distributed as dist
from sigsvgd_amd.utils.synthetic import synthetic_inputs
from sigsvgd_amd.distributed import ShardedSigSVGD
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29633")
dev = torch.device("cuda:0")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
X, s = synthetic_inputs(1024, 64, 7); X, s = X.to(dev), s.to(dev)
sh = ShardedSigSVGD(1.0, 1e-3)
for _ in range(3): sh.step(X, s)
acc = {}
for _ in range(10):
    sh.step(X, s, profile=True)
    for k, v in sh.phase_ms.items(): acc[k] = acc.get(k, 0) + v / 10
print({k: round(v, 4) for k, v in acc.items()})
torch.cuda.synchronize(); t0 = time.time()
for _ in range(50): Xn = sh.step(X, s)
torch.cuda.synchronize(); print("ms/step %.3f" % ((time.time() - t0) / 50 * 1e3))
dist.destroy_process_group()
