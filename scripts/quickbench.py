import sys, time, torch
sys.path.insert(0,'.')
from oracle import sigkernel_oracle as O
from sigsvgd_amd import ops
dev=torch.device('cuda:0')
for (N,T,d) in [(1024,64,7),(512,64,3),(128,32,7)]:
    X,s=O.synthetic_inputs(N,T,d); X=X.to(dev); s=s.to(dev)
    for name,fn in [('sym fwd+bwd',lambda: ops.gram_fwd_bwd(X,X,1.0,y_is_x=True)),('ordered fwd+bwd',lambda: ops.gram_fwd_bwd(X,X,1.0)),('fwd only',lambda: ops.gram_fwd(X,X,1.0))]:
        for _ in range(2): fn()
        torch.cuda.synchronize(); t=time.time()
        for _ in range(5): fn()
        torch.cuda.synchronize(); dt=(time.time()-t)/5
        print(f'N={N} T={T} d={d} {name}: {dt*1e3:.3f} ms', flush=True)
