import sys, torch
sys.path.insert(0,'.')
import bench
from gnn_amd.synth import CONFIGS, config_batch
import gnn_amd.function as Fn
dev=torch.device('cuda:0')
name=sys.argv[1]; sls=[int(v) for v in sys.argv[2].split(',')]
hp=dict(CONFIGS[name]['hp']); b=config_batch(name,1); bt=b.to_torch(dev)
for p in (0.2,0.0):
    for sl in sls:
        hp['dropout']=p
        Fn.choose_stream_len=lambda n,h,sl=sl: sl
        r=bench.time_kernels(None,bt,hp,iters=300)
        print(name,'p',p,'stream_len',sl,{k:v['us'] for k,v in r.items()}, flush=True)
