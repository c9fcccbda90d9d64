import os, sys, time, torch
sys.path.insert(0,'.')
from sigsvgd_amd.utils.synthetic import synthetic_inputs
from sigsvgd_amd import ops
dev=torch.device('cuda:0')
def t(fn,n=20):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0=time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time()-t0)/n*1e3
arg=sys.argv[1] if len(sys.argv)>1 else ''
shapes={'c4':[(1024,64,7)],'c5':[(256,128,14)],'stream':[(256,128,14),(256,100,7),(256,96,3)],'c5big':[(1024,128,14)],'short':[(1024,64,7),(128,32,7),(1024,32,7),(512,20,2)]}.get(arg,[(1024,64,7),(512,64,3),(128,32,7)])
sf = arg.endswith('q')
shapes = {'c5q':[(256,128,14)],'streamq':[(256,128,14),(256,100,7),(256,96,3)],'c5bigq':[(1024,128,14)]}.get(arg, shapes)
for (N,T,d) in shapes:
    X,s=synthetic_inputs(N,T,d); X=X.to(dev)
    print(f'N={N} T={T} d={d}: sym %.3f ms | ordered %.3f ms | fwd %.3f ms | fwd sym %.3f ms'%(t(lambda: ops.gram_fwd_bwd(X,X,1.0,y_is_x=True,stored_forward=sf)), t(lambda: ops.gram_fwd_bwd(X,X,1.0,stored_forward=sf)), t(lambda: ops.gram_fwd(X,X,1.0,stored_forward=sf)), t(lambda: ops.gram_fwd(X,X,1.0,y_is_x=True,stored_forward=sf))), flush=True)
