import sys, torch, json
sys.path.insert(0,'.')
import bench
from gnn_amd.synth import CONFIGS, config_batch
import gnn_amd.function as Fn
dev=torch.device('cuda:0')
name=sys.argv[1] if len(sys.argv)>1 else 'proteins1024'
hp=dict(CONFIGS[name]['hp']); b=config_batch(name,1); bt=b.to_torch(dev)
sls=[int(v) for v in sys.argv[2].split(',')] if len(sys.argv)>2 else [16]
for it in (20,200,2000):
  for p in (0.2,0.0):
    for sl in sls:
        hp['dropout']=p
        Fn.choose_stream_len=lambda n,h,sl=sl: sl
        r=bench.time_kernels(None,bt,hp,iters=it)
        print(name,'iters',it,'p',p,'stream_len',sl,{k:v['us'] for k,v in r.items()}, flush=True)
