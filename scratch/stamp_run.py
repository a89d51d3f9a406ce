import sys, ctypes, numpy as np, torch
sys.path.insert(0, '.')
from gnn_amd import _lib
_lib.LIB_PATH = 'scratch/librwk_stamp.so'
import bench
from gnn_amd.synth import CONFIGS, config_batch
dev = torch.device('cuda:0')
hp = dict(CONFIGS['stream_h64']['hp']); hp['dropout'] = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
bt = config_batch('stream_h64', 1).to_torch(dev)
r = bench.time_kernels(bt, hp, only=['fused_backward'])
print({k: v for k, v in r['fused_backward'].items() if k in ('us', 'TFLOP/s')})
lib = _lib.load()
buf = (ctypes.c_uint64 * 64)()
lib.rwk_debug_read_stamps.restype = ctypes.c_int
print('rc', lib.rwk_debug_read_stamps(buf))
a = np.array(list(buf), dtype=np.float64).reshape(8, 8)
names = ['B-store..top', 'stage', 'S1 wait', 'prefetch issue', 'phase A', 'S2 wait', 'pin(vm wait)', 'phase B mfma']
steps = 13
print('cycles per step, per wavefront (rows) and segment')
print(' ' * 8 + ' '.join(f'{n[:12]:>13s}' for n in names))
for w in range(8):
    print(f'wave {w}: ' + ' '.join(f'{a[w, k] / steps:13.0f}' for k in range(8)), ' total', round(a[w].sum() / steps))
