import sys, time, torch
sys.path.insert(0,'.')
from sigsvgd_amd.utils.synthetic import synthetic_inputs
from sigsvgd_amd import ops
dev=torch.device('cuda:0')
def t(fn,n=3):
    fn(); torch.cuda.synchronize(); t0=time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time()-t0)/n*1e3
for name,(N,T,d,n) in {'C1':(16,20,2,2),'C2':(128,32,7,0),'C3':(512,64,3,0),'C4':(1024,64,7,0),'C5/16 (N=256)':(256,128,14,0),'ref obstacle (20,5,2,n5)':(20,5,2,5),'ref robot (20,3,7,n6)':(20,3,7,6),'ref maze (35,30,2,n3)':(35,30,2,3)}.items():
    X,s=synthetic_inputs(N,T,d); X=X.to(dev); s=s.to(dev)
    def it():
        K,g=ops.gram_fwd_bwd(X,X,1.0,n,y_is_x=True); ops.svgd_phi(K,s,g,X=X,lr=1e-3)
    print(f'{name}: N={N} T={T} d={d} n={n}: {t(it):.3f} ms/iter', flush=True)
