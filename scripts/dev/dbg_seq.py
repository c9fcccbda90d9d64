import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import c_oracle as C
from sigsvgd_amd import ops
def paths(A, T, d, seed, scale=0.05):
    rng = np.random.default_rng(seed)
    return np.cumsum(scale * rng.standard_normal((A, T, d)), axis=1).astype(np.float32)
def check(A, B, T, d, tag=""):
    X, Y = paths(A, T, d, 21), paths(B, T, d, 22)
    Kr, gr = C.gram_fwd_bwd(X, Y, 0.8, 0)
    K, g = ops.gram_fwd_bwd(torch.as_tensor(X).cuda(), torch.as_tensor(Y).cuda(), 1 / 0.8)
    print(tag, (A, B, T, d), "nonsym grad err %.1e" % (np.abs(g.cpu().numpy() - gr).max() / np.abs(gr).max()), flush=True)
def poison(A, T, d, sym):
    X = torch.as_tensor(paths(A, T, d, 21)).cuda()
    ops.gram_fwd_bwd(X, X, 1 / 0.8, y_is_x=sym); torch.cuda.synchronize()
mode = sys.argv[1]
if mode == "a": poison(1, 64, 7, False)
if mode == "b": poison(1, 64, 7, True)
if mode == "c": poison(1, 64, 3, True)
if mode == "d": poison(64, 64, 7, True)
if mode == "e": poison(1, 17, 2, True)
check(1, 9, 17, 2, mode)
check(1, 9, 17, 7, mode)
check(3, 9, 40, 3, mode)
# does stale workspace content matter?
for key, ws in ops._WS.items():
    ws.fill_(0x3f)
torch.cuda.synchronize()
check(1, 9, 17, 2, mode + " ws=0x3f")
for key, ws in ops._WS.items():
    ws.zero_()
torch.cuda.synchronize()
check(1, 9, 17, 2, mode + " ws=0")
