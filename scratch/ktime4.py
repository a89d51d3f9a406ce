import sys, torch
sys.path.insert(0,'.')
import bench
from gnn_amd.synth import CONFIGS, config_batch
dev=torch.device('cuda:0')
for name in sys.argv[1:]:
    hp=dict(CONFIGS[name]['hp']); b=config_batch(name,1); bt=b.to_torch(dev)
    print(name, b.n_graphs, b.n_nodes, b.nnz, hp)
    for p in (0.2,0.0):
        hp['dropout']=p
        r=bench.time_kernels(None,bt,hp,iters=20)
        print(name,'p',p,{k:(v['us'],v['TFLOP/s'],v['GB/s']) for k,v in r.items()}, flush=True)
