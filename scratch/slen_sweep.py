"""Diagnostic: fused forward/backward time against the stream length (resident-slot fill) on proteins1024."""
import sys, torch
sys.path.insert(0, '.')
import bench
import gnn_amd.function as fn
from gnn_amd.synth import CONFIGS, config_batch
dev = torch.device('cuda:0')
hp = dict(CONFIGS['proteins1024']['hp'])
bt = config_batch('proteins1024', 1).to_torch(dev)
orig = fn.choose_stream_len
for lf, lb in ((19, 26), (16, 16), (13, 13), (12, 12), (11, 11), (10, 10), (9, 9), (8, 8), (7, 7), (6, 6), (5, 5), (4, 4), (32, 16), (48, 48)):
    fn.choose_stream_len = lambda n, d, h, s, steps, backward=False: (lb if backward else lf)
    r = bench.time_kernels(bt, hp, only=['fused_forward', 'fused_backward'])
    n = 307842
    print(f"fwd len {lf:3d} wgs {-(-(-(-n // lf)) // 16):5d} us {r['fused_forward']['us']:7.2f} | "
          f"bwd len {lb:3d} wgs {-(-(-(-n // lb)) // 16):5d} us {r['fused_backward']['us']:7.2f}", flush=True)
