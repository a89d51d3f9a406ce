"""Evaluation forward (one launch) on a COLD cache: a 512 MB fill between launches pushes the batch out of the L2s and
the Infinity Cache, then ONE forward is timed with events; median of 30.  python scratch/eval_cold.py [config]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnn_amd import RW_NN
from gnn_amd.synth import CONFIGS, config_batch
dev = torch.device("cuda:0")
cfg = sys.argv[1] if len(sys.argv) > 1 else "proteins1024"
hp = CONFIGS[cfg]["hp"]
adj, feats, gi, y = config_batch(cfg, 1).to_torch(dev)
torch.manual_seed(0)
m = RW_NN(hp["input_dim"], hp["max_step"], hp["hidden_graphs"], hp["size_hidden_graphs"], hp["hidden_dim"],
          hp["penultimate_dim"], hp["normalize"], hp["n_classes"], hp["dropout"], dev).to(dev).eval()
junk = torch.empty(128 * 1024 * 1024, dtype=torch.float32, device=dev)
with torch.no_grad():
    for _ in range(5):
        m(adj, feats, gi)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = m(adj, feats, gi)
    times = []
    for _ in range(30):
        junk.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1) * 1e3)
times.sort()
print(cfg, "evaluation forward, one launch, cold caches: median %.1f us (min %.1f)" % (times[len(times) // 2], times[0]))
