import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sigsvgd_amd.utils.synthetic import synthetic_inputs
from sigsvgd_amd import ops
dev=torch.device('cuda:0')
X,s=synthetic_inputs(1024,64,7); X=X.to(dev)
for _ in range(int(sys.argv[1]) if len(sys.argv)>1 else 3):
    ops.gram_fwd_bwd(X,X,1.0,y_is_x=True)
torch.cuda.synchronize()
