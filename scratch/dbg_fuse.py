import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from _golden import GoldenCase, rel_err
import gnn_amd.function as fn
from gnn_amd import RW_NN
from oracle import rwnn_oracle as O
dev = torch.device('cuda:0')
for name in ("mutag_b0", "mutag_b2_norm", "tiny_graphs_step4"):
    c = GoldenCase(name); hp = c.hp
    res = {}
    for fuse in (True, False):
        fn.FUSED_STEP_LAUNCHES = fuse
        m = RW_NN(hp["input_dim"], hp["max_step"], hp["hidden_graphs"], hp["size_hidden_graphs"], hp["hidden_dim"],
                  hp["penultimate_dim"], bool(hp["normalize"]), hp["n_classes"], 0.0, dev)
        m.load_state_dict(c.torch_state()); m = m.to(dev).eval()
        adj, feats, gi, y = c.torch_inputs(dev)
        feats = feats.clone().requires_grad_(True)
        k = m.rw_features(adj, feats, gi)
        k.backward(torch.from_numpy(c["dk"]).to(dev))
        res[fuse] = dict(k=k.detach().cpu().numpy(), fcb=m.fc.bias.grad.cpu().numpy(), fcw=m.fc.weight.grad.cpu().numpy(),
                         fh=m.features_hidden.grad.cpu().numpy(), feat=feats.grad.cpu().numpy(),
                         ah=None if m.adj_hidden.grad is None else m.adj_hidden.grad.cpu().numpy())
    w = c.params["fc.weight"].astype(np.float64); b = c.params["fc.bias"].astype(np.float64)
    x0 = 1.0 / (1.0 + np.exp(-(c["features"].astype(np.float64) @ w.T + b)))
    csr = (c["csr_crow"], c["csr_col"], c["csr_val"])
    d_x0, d_adj, d_feat = O.rw_kernel_backward(csr, x0, c.params["adj_hidden"], c.params["features_hidden"], c["graph_indicator"], hp["max_step"], c["dk"], hp["normalize"])
    d_pre = d_x0 * x0 * (1 - x0)
    want = dict(fcb=d_pre.sum(0), fcw=d_pre.T @ c["features"].astype(np.float64), fh=d_feat, feat=d_pre @ w)
    print(name, 'terms scale', np.abs(d_pre).sum(0), 'sum', d_pre.sum(0))
    for key in want:
        print('  ', key, 'fused err %.3e  separate err %.3e  ref-own err %.3e' % (rel_err(res[True][key], want[key]), rel_err(res[False][key], want[key]), rel_err(c.grads[{'fcb':'fc.bias','fcw':'fc.weight','fh':'features_hidden','feat':'features'}[key]], want[key])),
              'bit-equal' if np.array_equal(res[True][key], res[False][key]) else 'differs')
    print('   k bit-equal', np.array_equal(res[True]['k'], res[False]['k']))
