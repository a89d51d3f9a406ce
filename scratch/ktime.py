import sys, torch, json
sys.path.insert(0,'.')
import bench
from gnn_amd.synth import CONFIGS, config_batch
from gnn_amd import RW_NN
import gnn_amd.function as Fn
dev=torch.device('cuda:0')
name=sys.argv[1] if len(sys.argv)>1 else 'proteins1024'
hp=dict(CONFIGS[name]['hp']); b=config_batch(name,1); bt=b.to_torch(dev)
for p in (0.2,0.0):
    for sl in (None,8,16,32,64):
        hp['dropout']=p
        if sl is not None:
            Fn.choose_stream_len=lambda n,h,sl=sl: sl
            bench.__dict__  # time_kernels imports choose_stream_len at call time from gnn_amd.function
        r=bench.time_kernels(None,bt,hp,iters=20)
        print(name,'p',p,'stream_len',sl,{k:v['us'] for k,v in r.items()})
