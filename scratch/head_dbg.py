import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, torch, torch.nn.functional as F
import importlib.util
spec=importlib.util.spec_from_file_location('th','tests/test_gpu_head.py'); th=importlib.util.module_from_spec(spec); spec.loader.exec_module(th)
from gnn_amd.head import fused_head
dev=torch.device('cuda:0')
for (g,c,p,nc) in [(64,32,32,2),(1000,128,32,7),(8192,48,20,3),(8193,48,20,3),(9000,48,20,3)]:
    m=th.make(c,p,nc,g+c); m64=th.make(c,p,nc,g+c).double()
    rng=np.random.default_rng(g)
    k=torch.from_numpy((rng.uniform(0,1,size=(g,c))*3+50).astype(np.float32)); y=torch.from_numpy(rng.integers(0,nc,size=g))
    k64=k.double().requires_grad_(True); o,l=th.reference(m64,k64,y,True); l.backward()
    md=m.to(dev).train(); kd=k.to(dev).requires_grad_(True)
    out=fused_head(kd,md.bn,md.fc1,md.fc2,0.0,True); F.cross_entropy(out,y.to(dev)).backward()
    print(g,c,p,nc,'out',th.rel(out,o),'dK',th.rel(kd.grad,k64.grad))
    for (n,a),(_,b) in zip(md.named_parameters(), m64.named_parameters()): print('   ',n, th.rel(a.grad,b.grad))
    err=(kd.grad.cpu().double()-k64.grad).abs()
    print('   worst rows', torch.topk(err.max(dim=1).values,5).indices.tolist())
