import sys, time, torch
sys.path.insert(0,'.')
from sigsvgd_amd import ops
dev=torch.device('cuda:0')
for N,D in [(1024,448),(4096,1792),(512,192)]:
    K=torch.randn(N,N,device=dev); s=torch.randn(N,D,device=dev); g=torch.randn(N,D,device=dev)
    for _ in range(5): ops.svgd_phi(K,s,g)
    torch.cuda.synchronize(); t0=time.time()
    for _ in range(50): ops.svgd_phi(K,s,g)
    torch.cuda.synchronize(); dt=(time.time()-t0)/50
    print(f'N={N} D={D}: {dt*1e6:.1f} us  {2*N*N*D/dt/1e12:.2f} TFLOP/s', flush=True)
