"""Host-side profile of the eager drop-in step (model(...), cross_entropy, backward, gnn_amd.Adam.step)."""
import cProfile, pstats, sys, time, io
import torch, torch.nn.functional as F
sys.path.insert(0, ".")
import gnn_amd
from gnn_amd import RW_NN
from gnn_amd.synth import CONFIGS, config_batch

dev = torch.device("cuda:0")
cfg = CONFIGS["proteins1024"]; hp = cfg["hp"]
adj, feats, gi, y = config_batch("proteins1024", seed=1).to_torch(dev)
torch.manual_seed(0)
m = RW_NN(hp["input_dim"], hp["max_step"], hp["hidden_graphs"], hp["size_hidden_graphs"], hp["hidden_dim"],
          hp["penultimate_dim"], hp["normalize"], hp["n_classes"], hp["dropout"], dev).to(dev).train()
opt = gnn_amd.Adam(m.parameters(), lr=1e-3)

def step():
    opt.zero_grad()
    out = m(adj, feats, gi)
    loss = F.cross_entropy(out, y)
    loss.backward()
    opt.step()

import gnn_amd.composite as comp
for on in (True, False, True, False):
    comp.COMPOSITE_STEP = on
    for _ in range(50): step()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(300): step()
    torch.cuda.synchronize()
    print("composite", on, "ms/step", (time.perf_counter() - t) / 300 * 1e3)
comp.COMPOSITE_STEP = True
# phases, host time only
ph = [0.0] * 5
for _ in range(300):
    t0 = time.perf_counter(); opt.zero_grad()
    t1 = time.perf_counter(); out = m(adj, feats, gi)
    t2 = time.perf_counter(); loss = F.cross_entropy(out, y)
    t3 = time.perf_counter(); loss.backward()
    t4 = time.perf_counter(); opt.step()
    t5 = time.perf_counter()
    for i, (a, b) in enumerate(((t0, t1), (t1, t2), (t2, t3), (t3, t4), (t4, t5))): ph[i] += b - a
torch.cuda.synchronize()
print("host us: zero_grad %.1f forward %.1f loss %.1f backward %.1f opt %.1f" % tuple(p / 300 * 1e6 for p in ph))
pr = cProfile.Profile(); pr.enable()
for _ in range(300): step()
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(45); print(s.getvalue()[:9000])
