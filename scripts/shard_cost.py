import sys, time, torch
sys.path.insert(0,'.')
from sigsvgd_amd.utils.synthetic import synthetic_inputs
from sigsvgd_amd import ops
dev=torch.device('cuda:0')
X,s=synthetic_inputs(1024,64,7); X=X.to(dev); s=s.to(dev)
def t(fn,n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0=time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time()-t0)/n*1e3
for G in (1,2,4,8):
    def step():
        Kp,gp=ops.gram_sym_partial(X,1.0,0,G)
        v=ops.svgd_phi(Kp,s,gp.to(s.dtype))
        return v
    def part(): ops.gram_sym_partial(X,1.0,0,G)
    Kp,gp=ops.gram_sym_partial(X,1.0,0,G)
    print(f'G={G}: partial+phi {t(step):.3f} ms | partial only {t(part):.3f} ms | phi only {t(lambda: ops.svgd_phi(Kp,s,s)):.3f} ms | worst rank partial {max(t(lambda r=r: ops.gram_sym_partial(X,1.0,r,G),5) for r in range(G)):.3f}', flush=True)
