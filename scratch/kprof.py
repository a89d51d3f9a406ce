import sys, torch
sys.path.insert(0,'.')
import bench
from gnn_amd.synth import CONFIGS, config_batch
dev=torch.device('cuda:0')
name=sys.argv[1] if len(sys.argv)>1 else 'proteins1024'
p=float(sys.argv[2]) if len(sys.argv)>2 else 0.2
hp=dict(CONFIGS[name]['hp']); hp['dropout']=p
b=config_batch(name,1); bt=b.to_torch(dev)
r=bench.time_kernels(None,bt,hp,iters=5)
print(r)
