"""Diagnostic: cycle shares of the head kernels (library built with -DRWK_STAMP as scratch/librwk_stamp.so)."""
import sys, ctypes, numpy as np, torch
sys.path.insert(0, '.')
from gnn_amd import _lib
_lib.LIB_PATH = 'scratch/librwk_stamp.so'
from gnn_amd import RW_NN
from gnn_amd.synth import CONFIGS, config_batch
from gnn_amd.trainer import Trainer
dev = torch.device('cuda:0')
hp = dict(CONFIGS['proteins1024']['hp'])
b = config_batch('proteins1024', 1)
bt = b.to_torch(dev)
m = RW_NN(hp["input_dim"], hp["max_step"], hp["hidden_graphs"], hp["size_hidden_graphs"], hp["hidden_dim"],
          hp["penultimate_dim"], bool(hp["normalize"]), hp["n_classes"], hp["dropout"], dev).to(dev)
m.train()
tr = Trainer(m, bt, 1)
for _ in range(5):
    tr.step_eager()
torch.cuda.synchronize()
lib = _lib.load()
buf = (ctypes.c_uint64 * 32)()
lib.rwk_debug_read_head_stamps.restype = ctypes.c_int
print('rc', lib.rwk_debug_read_head_stamps(buf))
a = list(buf)
print('apply   : stats+params %d | s_y %d | fc1+dropout %d | fc2 %d | log-softmax %d' % tuple(a[0:5]))
print('bwd_tile: loads+stores %d | d pre-act %d | dY %d | param grads %d | write partials %d' % tuple(a[8:13]))
print('train   : params+stats %d | normalise %d | fc1+dropout %d | fc2 %d | log-softmax+loss %d | d pre-act %d | dY %d | param grads %d | stores %d'
      % tuple(a[16:25]))
for _ in range(20):
    tr.step_eager()
torch.cuda.synchronize()
lib.rwk_debug_read_head_stamps(buf)
a = list(buf)
print('train   : params+stats %d | normalise %d | fc1+dropout %d | fc2 %d | log-softmax+loss %d | d pre-act %d | dY %d | param grads %d | stores %d'
      % tuple(a[16:25]))
print('train prologue: issue param loads %d | issue stats loads %d | params to LDS %d | merge %d | finish cols %d | barrier->%d' % tuple(a[25:30] + [a[16]]))
print('train32: issue loads %d | stats loads back + block sums %d | merge %d | setup %d | BN + fc1 products %d | reduce-scatter, relu, dropout %d | fc2 %d | softmax, loss, dlogits %d | d pre-act, dY %d | dY store, LDS staging %d | parameter sums %d | stores %d' % tuple(a[16:28]))
