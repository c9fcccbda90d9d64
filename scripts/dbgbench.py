import os, sys, time, torch
sys.path.insert(0,'.')
from oracle import sigkernel_oracle as O
from sigsvgd_amd import ops
dev=torch.device('cuda:0')
X,s=O.synthetic_inputs(1024,64,7); X=X.to(dev)
def t(fn,n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0=time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time()-t0)/n*1e3
for dbg in sys.argv[1:]:
    os.environ['SIGSVGD_DBG']=dbg
    print('dbg',dbg,'sym %.3f ms'%t(lambda: ops.gram_fwd_bwd(X,X,1.0,y_is_x=True)), flush=True)
os.environ['SIGSVGD_DBG']='0'
print('ordered %.3f ms'%t(lambda: ops.gram_fwd_bwd(X,X,1.0)))
