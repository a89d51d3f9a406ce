"""Diagnostic: forward + backward time of gnn_amd.rw_gnn.RW_GNN on the proteins1024 batch (eager launches)."""
import sys, time, torch
sys.path.insert(0, '.')
from gnn_amd.rw_gnn import RW_GNN
from gnn_amd.synth import CONFIGS, config_batch
dev = torch.device('cuda:0')
hp = CONFIGS['proteins1024']['hp']
b = config_batch('proteins1024', 1)
adj, feats, gi, y = b.to_torch(dev)
ei = adj._indices().contiguous()
torch.manual_seed(0)
m = RW_GNN(hp['input_dim'], hp['max_step'], hp['hidden_graphs'], hp['size_hidden_graphs'], hp['hidden_dim'],
           hp['penultimate_dim'], False, hp['n_classes'], hp['dropout'], dev).to(dev).train()
m.seed_dropout(1)
data = RW_GNN.Data(feats, ei, gi)
opt = None
def step():
    global opt
    out = m(data)
    if opt is None:
        opt = torch.optim.Adam(m.parameters(), lr=1e-2)
    opt.zero_grad(set_to_none=True)
    torch.nn.functional.cross_entropy(out, y).backward()
    opt.step()
for _ in range(10): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50): step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 50
print('RW_GNN train step: %.3f ms, %.2f M graphs/s (1024 graphs, %d nodes)' % (dt * 1e3, 1024 / dt / 1e6, b.n_nodes))
from gnn_amd.trainer import Trainer
tr = Trainer(m, (adj, feats, gi, y), 1, forward_fn=lambda: m(data))
tr.capture()
for _ in range(10): tr.step_graph()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(100): tr.step_graph()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 100
print('RW_GNN train step, graph replay: %.3f ms, %.2f M graphs/s, loss %.4f' % (dt * 1e3, 1024 / dt / 1e6, float(tr.loss)))
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(5): step()
    torch.cuda.synchronize()
rows = sorted(prof.key_averages(), key=lambda e: -e.device_time_total)[:8]
for e in rows:
    print('%-70s %4d %8.1f us' % (e.key[:70], e.count, e.device_time_total / 5))
