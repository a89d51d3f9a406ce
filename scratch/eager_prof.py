import sys, cProfile, pstats, io, time, torch
sys.path.insert(0, '.')
import bench
from gnn_amd import RW_NN
from gnn_amd.synth import CONFIGS, config_batch
dev = torch.device('cuda:0')
hp = CONFIGS['proteins1024']['hp']
bt = config_batch('proteins1024', 1).to_torch(dev)
torch.manual_seed(0)
m = RW_NN(hp['input_dim'], hp['max_step'], hp['hidden_graphs'], hp['size_hidden_graphs'], hp['hidden_dim'],
          hp['penultimate_dim'], hp['normalize'], hp['n_classes'], hp['dropout'], dev).to(dev).train()
m.seed_dropout(1)
tr = bench.Trainer(m, bt, 1)
for _ in range(30): tr.step_eager()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200): tr.step_eager()
torch.cuda.synchronize()
print('eager us/step', (time.perf_counter() - t0) / 200 * 1e6)
pr = cProfile.Profile(); pr.enable()
for _ in range(200): tr.step_eager()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(22); print(s.getvalue()[:4500])
