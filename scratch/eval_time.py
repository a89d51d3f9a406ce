"""Evaluation-mode forward (rwgnn2/main.py:110-117) of the proteins1024 batch: per-kernel times of the Gram path against
the stream kernels with the identity mask."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gnn_amd.function as fn
from gnn_amd import RW_NN
from gnn_amd.synth import CONFIGS, config_batch
dev = torch.device("cuda:0")
cfg = sys.argv[1] if len(sys.argv) > 1 else "proteins1024"
hp = CONFIGS[cfg]["hp"]
b = config_batch(cfg, 1)
adj, feats, gi, y = b.to_torch(dev)
torch.manual_seed(0)
m = RW_NN(hp["input_dim"], hp["max_step"], hp["hidden_graphs"], hp["size_hidden_graphs"], hp["hidden_dim"],
          hp["penultimate_dim"], hp["normalize"], hp["n_classes"], hp["dropout"], dev).to(dev).eval()
import gnn_amd.eval as ev
for name, fused, gram in (("one launch", True, True), ("separate launches, Gram path", False, True),
                          ("separate launches, stream kernels with the identity mask", False, False)):
    ev.FUSED_EVAL_FORWARD = fused
    fn.GRAM_FAST_PATH = gram
    with torch.no_grad():
        for _ in range(5):
            m(adj, feats, gi)
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            m(adj, feats, gi)
        torch.cuda.current_stream().wait_stream(s)
        with torch.cuda.graph(g):
            for _ in range(10):
                out = m(adj, feats, gi)
        for _ in range(5):
            g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        print(cfg, "evaluation forward of the whole model, %s: %.2f us" % (name, e0.elapsed_time(e1) * 1e3 / 200))
