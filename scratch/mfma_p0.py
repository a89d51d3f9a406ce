"""Matrix-core kernels of stream_h64 with and without dropout (the generator's share)."""
import sys, torch
sys.path.insert(0, '.')
import bench
from gnn_amd.synth import CONFIGS, config_batch
dev = torch.device('cuda:0')
bt = config_batch('stream_h64', 1).to_torch(dev)
for p in (0.2, 0.0):
    hp = dict(CONFIGS['stream_h64']['hp']); hp['dropout'] = p
    r = bench.time_kernels(bt, hp, only=['fused_forward', 'fused_backward'])
    print('dropout', p, {k: (round(v['us'], 1), round(v['TFLOP/s_executed'], 1)) for k, v in r.items() if 'fused' in k})
