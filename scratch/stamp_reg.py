"""Diagnostic: cycle shares of the D<=4 backward kernel (library built with -DRWK_STAMP as scratch/librwk_stamp.so)."""
import sys, ctypes, numpy as np, torch
sys.path.insert(0, '.')
from gnn_amd import _lib
_lib.LIB_PATH = 'scratch/librwk_stamp.so'
import bench
from gnn_amd.synth import CONFIGS, config_batch
dev = torch.device('cuda:0')
for drop in (0.2, 0.0):
    hp = dict(CONFIGS['proteins1024']['hp']); hp['dropout'] = drop
    bt = config_batch('proteins1024', 1).to_torch(dev)
    r = bench.time_kernels(bt, hp, only=['fused_backward'])
    print('dropout', drop, {k: v for k, v in r['fused_backward'].items() if k in ('us', 'TFLOP/s')}, r.get('shape'))
    lib = _lib.load()
    buf = (ctypes.c_uint64 * 32)()
    lib.rwk_debug_read_reg_stamps.restype = ctypes.c_int
    print('rc', lib.rwk_debug_read_reg_stamps(buf))
    a = np.array(list(buf), dtype=np.float64).reshape(4, 8)
    names = ['prolog/store', 'tile load', 'inner loop', 'to epilogue', 'epilogue', '-', '-', 'life 10ns']
    print(' ' * 8 + ' '.join(f'{n[:12]:>13s}' for n in names))
    for w in range(4):
        print(f'wave {w}: ' + ' '.join(f'{a[w, k]:13.0f}' for k in range(8)), ' cycles', round(a[w, :5].sum()),
              ' GHz', round(a[w, :5].sum() / max(a[w, 7], 1) / 10, 3))
    nb = 741
    bb = (ctypes.c_uint64 * 4096)()
    lib.rwk_debug_read_reg_blocks.restype = ctypes.c_int
    lib.rwk_debug_read_reg_blocks(bb)
    t = np.array(list(bb), dtype=np.float64).reshape(2048, 2)[:nb] / 100.0   # microseconds
    t0 = t[:, 0].min()
    st, en = t[:, 0] - t0, t[:, 1] - t0
    print('blocks', nb, 'start us: min %.2f p50 %.2f p90 %.2f max %.2f | end us: min %.2f p50 %.2f max %.2f | life p50 %.2f max %.2f'
          % (st.min(), np.median(st), np.percentile(st, 90), st.max(), en.min(), np.median(en), en.max(),
             np.median(en - st), (en - st).max()))
    order = np.argsort(st)
    print('start by block id (every 64th):', ' '.join('%d:%.1f' % (i, st[i]) for i in range(0, nb, 64)))
    print('life by block id (every 64th):', ' '.join('%d:%.1f' % (i, en[i] - st[i]) for i in range(0, nb, 64)))
