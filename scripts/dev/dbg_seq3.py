import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import c_oracle as C
from sigsvgd_amd import ops, _lib
def paths(A, T, d, seed, scale=0.05):
    rng = np.random.default_rng(seed)
    return np.cumsum(scale * rng.standard_normal((A, T, d)), axis=1).astype(np.float32)
def use(path):
    _lib._lib = None
    _lib.LIB_PATH = os.path.abspath(path)
    _lib.load()
def check(tag):
    A, B, T, d = 1, 9, 17, 2
    X, Y = paths(A, T, d, 21), paths(B, T, d, 22)
    Kr, gr = C.gram_fwd_bwd(X, Y, 0.8, 0)
    K, g = ops.gram_fwd_bwd(torch.as_tensor(X).cuda(), torch.as_tensor(Y).cuda(), 1 / 0.8)
    print(tag, "nonsym grad err %.1e" % (np.abs(g.cpu().numpy() - gr).max() / np.abs(gr).max()), flush=True)
def poison():
    X = torch.as_tensor(paths(1, 64, 7, 21)).cuda()
    ops.gram_fwd_bwd(X, X, 1 / 0.8, y_is_x=True); torch.cuda.synchronize()
poison_lib, victim_lib = sys.argv[1], sys.argv[2]
use(poison_lib); poison()
use(victim_lib); check(f"poison={os.path.basename(poison_lib)} victim={os.path.basename(victim_lib)}")
