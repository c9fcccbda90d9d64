import sys, time, torch
sys.path.insert(0,'.')
from oracle import sigkernel_oracle as O
from sigsvgd_amd import ops
dev=torch.device('cuda:0')
def t(fn,n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0=time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time()-t0)/n*1e3
for name,(N,T,d,n) in {'notebook (100,10,2,n4)':(100,10,2,4),'maze (35,30,2,n3)':(35,30,2,3),'obstacle (20,5,2,n5)':(20,5,2,5),'robot (20,3,7,n6)':(20,3,7,6),'big (256,10,2,n4)':(256,10,2,4)}.items():
    X,s=O.synthetic_inputs(N,T,d); X=X.to(dev); s=s.to(dev)
    def it():
        K,g=ops.gram_fwd_bwd(X,X,1.0,n,y_is_x=True); ops.svgd_phi(K,s,g,X=X,lr=1e-3)
    print(f'{name}: {t(it):.3f} ms/iter', flush=True)
