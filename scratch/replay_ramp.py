"""Device time of each of the first replays of the captured training step (events between replays): how long the
step takes to reach its steady duration after capture, and after the device sat idle."""
import sys, time
import torch
sys.path.insert(0, '.')
from gnn_amd import RW_NN
from gnn_amd.synth import CONFIGS, config_batch
from gnn_amd.trainer import Trainer

dev = torch.device('cuda', 0)
hp = dict(CONFIGS['proteins1024']['hp'])
batch_t = config_batch('proteins1024', seed=1).to_torch(dev)
torch.manual_seed(0)
model = RW_NN(hp["input_dim"], hp["max_step"], hp["hidden_graphs"], hp["size_hidden_graphs"], hp["hidden_dim"],
              hp["penultimate_dim"], hp["normalize"], hp["n_classes"], hp["dropout"], dev).to(dev)
model.train(); model.seed_dropout(1234)
tr = Trainer(model, batch_t, 1, "flat")
tr.capture()

def ramp(n, label):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    torch.cuda.synchronize()
    ev[0].record()
    for i in range(n):
        tr.step_graph()
        ev[i + 1].record()
    torch.cuda.synchronize()
    d = [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(n)]
    print(label, ' '.join('%.0f' % v for v in d))

ramp(40, 'after capture:')
ramp(40, 'again:        ')
time.sleep(0.05)
ramp(40, 'after 50 ms idle:')
time.sleep(1.0)
ramp(40, 'after 1 s idle:')
