"""Host-side profile of the reference's unchanged training closure (rwgnn2/main.py:82-88) on gnn_amd.RW_NN."""
import cProfile, pstats, sys, time, io
import torch, torch.nn.functional as F
sys.path.insert(0, '.')
from gnn_amd import RW_NN
from gnn_amd.synth import CONFIGS, config_batch
dev = torch.device('cuda:0')
hp = dict(CONFIGS['proteins1024']['hp'])
adj, feats, gi, y = config_batch('proteins1024', 1).to_torch(dev)
torch.manual_seed(0)
model = RW_NN(hp["input_dim"], hp["max_step"], hp["hidden_graphs"], hp["size_hidden_graphs"], hp["hidden_dim"],
              hp["penultimate_dim"], hp["normalize"], hp["n_classes"], hp["dropout"], dev).to(dev)
model.train()
opt = torch.optim.Adam(model.parameters(), lr=1e-2)
def train():
    opt.zero_grad()
    out = model(adj, feats, gi)
    loss = F.cross_entropy(out, y)
    loss.backward()
    opt.step()
    return loss
for _ in range(30): train()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(300): train()
torch.cuda.synchronize()
print('drop-in step ms', (time.perf_counter() - t0) / 300 * 1e3)
pr = cProfile.Profile(); pr.enable()
for _ in range(300): train()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(28); print(s.getvalue()[:6000])
