import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import gnn_amd.function as fn
from _golden import GoldenCase, rel_err
import test_gpu_parity as T
dev = torch.device("cuda:0")
for name in ["mutag_b0", "mutag_b2_norm", "tiny_graphs_step4"]:
    c = GoldenCase(name)
    adj, feats, gi, _ = c.torch_inputs(dev)
    dk = torch.from_numpy(c["dk"]).to(dev)
    for gram in (True, False):
        fn.GRAM_FAST_PATH = gram
        runs = []
        for fused in (True, False):
            fn.FUSED_STEP_LAUNCHES = fused
            runs.append(T._run_layer(c, dev, adj, feats, gi, dk))
        fn.FUSED_STEP_LAUNCHES = True
        a, b = runs[0]["features"].cpu().numpy(), runs[1]["features"].cpu().numpy().astype(np.float64)
        print(name, "gram", gram, "features fused-vs-separate rel", rel_err(a, b), "max", np.abs(b).max(),
              {k: bool(torch.equal(runs[0][k], runs[1][k])) for k in runs[0]})
