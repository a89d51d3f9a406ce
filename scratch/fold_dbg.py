"""Bisect of a crash at the end of the capture of the step without the head's finishing launch."""
import os, sys, faulthandler
faulthandler.enable()
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
from _golden import GoldenCase
import bench
from gnn_amd import RW_NN, _lib
import gnn_amd.optim as optim
import gnn_amd.function as fn

variant = sys.argv[1]
dev = torch.device('cuda', 0)
c = GoldenCase("mutag_b0")
hp = dict(c.hp, dropout=0.2)
torch.manual_seed(0)
m = RW_NN(hp["input_dim"], hp["max_step"], hp["hidden_graphs"], hp["size_hidden_graphs"], hp["hidden_dim"],
          hp["penultimate_dim"], hp["normalize"], hp["n_classes"], hp["dropout"], dev).to(dev)
m.train(); m.seed_dropout(7)
tr = bench.Trainer(m, c.torch_inputs(dev), 1)
lib = _lib.load()
if variant == 'plain_bwd':      # the ordinary backward kernel reading dy as d_k (wrong values, legal accesses)
    real = lib.rwk_fused_backward
    def fake(*a):
        (x, z, gid, perm, counts, n, d, h, s, steps, sl, drop, norm, rng, dx, dzp, k, dy) = a[:18]
        return real(x, z, gid, perm, dy, counts, n, d, h, s, steps, sl, drop, norm, rng, dx, dzp, None, a[-1])
    class L:
        def __getattr__(self, name):
            return fake if name == 'rwk_fused_backward_bn' else getattr(lib, name)
    _lib.load = lambda: L()
    fn._lib.load = _lib.load
elif variant == 'plain_tail':
    orig = optim.StepTail.launch
    def launch(self, lib_, stream, head_srcs, *a, head_fold=None, n_graphs=0, head_columns=None):
        return orig(self, lib_, stream, head_srcs, *a)
    optim.StepTail.launch = launch
elif variant == 'nofold':
    fn.FOLD_HEAD_FINISH = False
for _ in range(2):
    tr.step_eager()
torch.cuda.synchronize()
print('eager ok', float(tr.loss), tr.tail.head_folds, flush=True)
tr.capture()
print('capture ok', flush=True)
tr.step_graph(); torch.cuda.synchronize()
print('replay ok', float(tr.loss), flush=True)
