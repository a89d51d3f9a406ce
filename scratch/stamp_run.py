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
names = ['phase A', 'stage', 'phase B', 'barrier', 'prefetch', 'per-block', 'kernel cyc', 'kernel 10ns']
steps = 26 * 4
print('pipelined backward: cycles per position step, per wavefront (rows) and segment; block 3 (first round)')
print(' ' * 8 + ' '.join(f'{n[:12]:>13s}' for n in names))
for w in range(8):
    print(f'wave {w}: ' + ' '.join(f'{a[w, k] / steps:13.0f}' for k in range(6)),
          f' kernel {a[w, 6]:.0f} cycles in {a[w, 7] / 100:.1f} us = {a[w, 6] / a[w, 7] / 10:.3f} GHz')
