"""Training step of a configuration with and without the per-graph chain launches (function.GRAPH_CHAIN).
Usage on the GPU box: PYTHONPATH=. python scratch/chain_ab.py powerlaw256"""
import sys
import torch
import gnn_amd.function as fn
from gnn_amd import RW_NN
from gnn_amd.synth import CONFIGS, config_batch
from gnn_amd.trainer import Trainer

name = sys.argv[1] if len(sys.argv) > 1 else "powerlaw256"
dev = torch.device("cuda:0")
hp = dict(CONFIGS[name]["hp"])
batch_t = config_batch(name, 1).to_torch(dev)
fn.GRAPH_CHAIN_MIN_GRAPHS = 1          # the comparison itself decides where the chain pays
for chain in (True, False, True, False):
    fn.GRAPH_CHAIN = chain
    torch.manual_seed(0)
    m = RW_NN(hp["input_dim"], hp["max_step"], hp["hidden_graphs"], hp["size_hidden_graphs"], hp["hidden_dim"],
              hp["penultimate_dim"], hp["normalize"], hp["n_classes"], hp["dropout"], dev).to(dev)
    m.train()
    m.seed_dropout(1)
    tr = Trainer(m, batch_t)
    tr.capture()
    for _ in range(20):
        tr.step_graph()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        tr.step_graph()
    e1.record()
    torch.cuda.synchronize()
    print(name, "chain" if chain else "spmm launches", "%.4f ms per step" % (e0.elapsed_time(e1) / 200),
          "loss %.5f" % float(tr.loss), flush=True)
