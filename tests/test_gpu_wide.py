"""Hyper-parameters beyond one launch of the fused kernels - more than 64 hidden graphs, hidden_dim above 64, more than
four steps (rwgnn2/main.py:22-27 accepts any) - through the composed path (gnn_amd/wide.py): K and every gradient
against the float64 oracle to 1e-5, without dropout; with dropout: the scale and the keep rate (statistically) and a
whole-model training step."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from _golden import rel_err
from oracle import rwnn_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def batch():
    from gnn_amd.synth import random_batch
    b = random_batch(23, 1, 30, 2.4, seed=77)
    return b.n_nodes, b.graph_indicator, b.indices, b.values


@pytest.mark.parametrize("h,s,d,steps,normalize", [
    (96, 4, 8, 2, False),        # hidden graphs in two chunks
    (16, 5, 128, 2, True),       # two channel tiles: four tile pairs per step
    (16, 3, 4, 6, False),        # six steps
    (70, 3, 72, 5, True),        # everything at once, ragged tiles (72 = 2 x 36)
    (130, 2, 3, 1, False),       # a single step, three chunks
])
def test_composed_layer_against_oracle(h, s, d, steps, normalize, batch, dev):
    from gnn_amd import get_structure, rw_kernel_layer
    n, gi, idx, val = batch
    rng = np.random.default_rng(h + s + d + steps)
    pre = rng.normal(size=(n, d)).astype(np.float32)
    adj_h = rng.uniform(-1, 1, size=(h, s * (s - 1) // 2)).astype(np.float32)
    z0 = (rng.uniform(0, 1, size=(h, s, d)) / np.sqrt(d)).astype(np.float32)
    g = int(gi.max()) + 1
    dk = rng.normal(size=(g, steps * h)).astype(np.float32)
    adj = torch.sparse_coo_tensor(torch.from_numpy(idx), torch.from_numpy(val), (n, n)).to(dev)
    st = get_structure(adj, torch.from_numpy(gi).to(dev))
    pre_t = torch.from_numpy(pre).to(dev).requires_grad_(True)
    adj_t = torch.from_numpy(adj_h).to(dev).requires_grad_(True)
    z_t = torch.from_numpy(z0).to(dev).requires_grad_(True)
    k = rw_kernel_layer(pre_t, adj_t, z_t, st, steps, normalize, 0.0, None)
    k.backward(torch.from_numpy(dk).to(dev))
    csr = O.coo_to_csr(idx, val, n)
    x0 = 1.0 / (1.0 + np.exp(-pre.astype(np.float64)))
    k_ref = O.rw_kernel_forward(csr, x0, adj_h, z0, gi, steps, normalize)
    d_x0, d_adj, d_feat = O.rw_kernel_backward(csr, x0, adj_h, z0, gi, steps, dk, normalize)
    assert k.shape == k_ref.shape == (g, steps * h)
    assert rel_err(k.detach().cpu().numpy(), k_ref) < TOL
    assert rel_err(z_t.grad.cpu().numpy(), d_feat) < TOL
    assert rel_err(pre_t.grad.cpu().numpy(), d_x0 * x0 * (1 - x0)) < TOL
    if steps > 1 and s > 1:
        assert rel_err(adj_t.grad.cpu().numpy(), d_adj) < TOL
    else:
        assert adj_t.grad is None or float(adj_t.grad.abs().max()) == 0.0


def test_composed_dropout_is_unbiased_and_advances(batch, dev):
    """With dropout the composed path keeps 1 - p of the products and scales by 1/(1-p): the mean over many draws
    approaches the no-dropout features, and consecutive calls draw different masks."""
    from gnn_amd import get_structure, rw_kernel_layer
    n, gi, idx, val = batch
    h, s, d, steps, p = 80, 3, 4, 2, 0.25
    rng = np.random.default_rng(1)
    pre = torch.from_numpy(rng.normal(size=(n, d)).astype(np.float32)).to(dev)
    adj_h = torch.from_numpy(rng.uniform(-1, 1, size=(h, 3)).astype(np.float32)).to(dev)
    z0 = torch.from_numpy(rng.uniform(0, 1, size=(h, s, d)).astype(np.float32)).to(dev)
    adj = torch.sparse_coo_tensor(torch.from_numpy(idx), torch.from_numpy(val), (n, n)).to(dev)
    st = get_structure(adj, torch.from_numpy(gi).to(dev))
    clean = rw_kernel_layer(pre, adj_h, z0, st, steps, False, 0.0, None)
    state = torch.tensor([4242, 0], dtype=torch.int64, device=dev)
    draws = [rw_kernel_layer(pre, adj_h, z0, st, steps, False, p, state) for _ in range(200)]
    assert int(state[1]) == 200
    assert not torch.equal(draws[0], draws[1])
    mean = torch.stack(draws).mean(0)
    assert float((mean - clean).abs().max() / clean.abs().max()) < 0.05
    # every (chunk, step) draws its own mask: the two chunks of hidden graphs are not copies of each other
    z_same = z0.clone()
    z_same[64:] = z0[:16]
    adj_same = adj_h.clone()
    adj_same[64:] = adj_h[:16]
    state.copy_(torch.tensor([4242, 0], device=dev))
    k = rw_kernel_layer(pre, adj_same, z_same, st, steps, False, p, state)
    assert not torch.equal(k[:, 64:80], k[:, 0:16])


def test_model_with_hyper_parameters_beyond_one_launch(dev):
    """RW_NN(hidden_graphs=96, hidden_dim=72, max_step=5): a training step with dropout runs and gives finite
    gradients for every parameter; evaluation-mode scores equal the float64 reference of the model's own operators."""
    from gnn_amd import RW_NN
    from gnn_amd.synth import random_batch
    b = random_batch(12, 4, 25, 2.5, seed=5, n_classes=3)
    adj, feats, gi, y = b.to_torch(dev)
    torch.manual_seed(0)
    m = RW_NN(1, 5, 96, 3, 72, 16, True, 3, 0.2, dev).to(dev).train()
    out = m(adj, feats, gi)
    F.cross_entropy(out, y).backward()
    assert out.shape == (12, 3) and bool(torch.isfinite(out).all())
    for name, p in m.named_parameters():
        assert p.grad is not None and bool(torch.isfinite(p.grad).all()), name
    m.eval()
    with torch.no_grad():
        got = m(adj, feats, gi).cpu().numpy()
    from oracle import rwnn_torch_port as P
    params = P.cast_params({k: v.detach().cpu() for k, v in m.state_dict().items()}, torch.float64)
    a64, f64, gi64, _ = b.to_torch("cpu")
    want, _ = P.forward(params, a64.coalesce().double(), f64.double(), gi64, 5, True, 0.2, False,
                        (m.bn.running_mean.detach().cpu().double(), m.bn.running_var.detach().cpu().double()))
    assert rel_err(got, want.detach().numpy()) < 1e-4
