"""RW_GNN (the torch_geometric variant, rwgnn/rwgnn_pyg.py:48-170) on the HIP kernels against the float64 torch
restatement in oracle/rwgnn_pyg_port.py: the per-step features kept per hidden node, their gradients with respect
to every parameter that feeds them, and the model's interface.  Parity is unpinned for this model (the reference
cannot run here: torch_geometric is absent)."""
import numpy as np
import pytest
import torch

from _golden import rel_err
from oracle import rwgnn_pyg_port as P

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _batch(seed, n_graphs, f):
    rng = np.random.default_rng(seed)
    sizes = rng.integers(2, 30, size=n_graphs)
    n = int(sizes.sum())
    batch = np.repeat(np.arange(n_graphs), sizes)
    src, dst, start = [], [], 0
    for m in sizes:
        e = rng.integers(0, m, size=(2 * m, 2))
        src += [start + e[:, 0], start + e[:, 1]]
        dst += [start + e[:, 1], start + e[:, 0]]
        start += m
    ei = np.vstack([np.concatenate(src), np.concatenate(dst)]).astype(np.int64)
    x = rng.normal(size=(n, f)).astype(np.float32)
    return x, ei, batch.astype(np.int64)


@pytest.mark.parametrize("h,s,d,steps,f", [(16, 5, 4, 2, 3), (4, 3, 4, 3, 1), (16, 5, 8, 2, 5), (10, 7, 2, 4, 2),
                                           (20, 4, 16, 1, 4)])
def test_rw_gnn_features_and_gradients(h, s, d, steps, f, dev):
    from gnn_amd.rw_gnn import RW_GNN
    x, ei, batch = _batch(h + s + d, 23, f)
    torch.manual_seed(h * 7 + s)
    m = RW_GNN(f, steps, h, s, d, 32, False, 3, 0.0, dev).to(dev)
    m.train()
    data = RW_GNN.Data(torch.from_numpy(x).to(dev), torch.from_numpy(ei).to(dev), torch.from_numpy(batch).to(dev))
    feats = m.rw_features(data)
    g = int(batch.max()) + 1
    assert feats.shape == (g, h * s * steps)
    rng = np.random.default_rng(5)
    d_out = rng.normal(size=tuple(feats.shape)).astype(np.float32)
    feats.backward(torch.from_numpy(d_out).to(dev))

    names = ["adj_hidden", "features_hidden", "fc.weight", "fc.bias"]
    names += [f"rw_layers.{i}.lin.{w}" for i in range(steps - 1) for w in ("weight", "bias")]
    sd = dict(m.named_parameters())
    params = {k: sd[k].detach().cpu().double().requires_grad_(True) for k in names}
    want = P.rw_features(params, torch.from_numpy(x).double(), torch.from_numpy(ei), torch.from_numpy(batch), steps)
    assert rel_err(feats.detach().cpu().numpy(), want.detach().numpy()) < TOL
    want.backward(torch.from_numpy(d_out).double())
    for k in names:
        if k == "adj_hidden" and steps == 1:
            assert sd[k].grad is None
            continue
        assert rel_err(sd[k].grad.cpu().numpy(), params[k].grad.numpy()) < TOL, k
    # the last RWConv layer is declared but never used (rwgnn_pyg.py:71-73,128): no gradient reaches it
    assert sd[f"rw_layers.{steps - 1}.lin.weight"].grad is None


def test_rw_gnn_interface_and_dropout(dev):
    from gnn_amd.rw_gnn import RW_GNN
    x, ei, batch = _batch(1, 40, 2)
    torch.manual_seed(0)
    m = RW_GNN(2, 2, 16, 5, 4, 32, False, 2, 0.2, dev).to(dev)
    assert [k for k, _ in m.named_parameters()][:4] == ["adj_hidden", "features_hidden", "fc.weight", "fc.bias"]
    assert m.bn is None and m.fc1 is None and m.fc2 is None
    data = RW_GNN.Data(torch.from_numpy(x).to(dev), torch.from_numpy(ei).to(dev), torch.from_numpy(batch).to(dev))
    m.train()
    m.seed_dropout(3)
    out = m(data)
    assert out.shape == (40, 2) and bool(torch.isfinite(out).all())
    assert m.bn.num_features == 16 * 5 * 2 and m.fc1.in_features == 160 and m.fc2.out_features == 2
    assert abs(float(out.exp().sum(1).mean()) - 1.0) < 1e-5
    torch.nn.functional.nll_loss(out, torch.zeros(40, dtype=torch.int64, device=dev)).backward()
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all())
               for k, p in m.named_parameters() if not k.startswith("rw_layers.1"))
    # same generator state, same masks; the next call draws new ones
    m.seed_dropout(3)
    a = m.rw_features(data)
    m.seed_dropout(3)
    b = m.rw_features(data)
    c = m.rw_features(data)
    assert torch.equal(a, b) and not torch.equal(b, c)
    m.eval()
    e1, e2 = m.rw_features(data), m.rw_features(data)
    assert torch.equal(e1, e2)
    # normalize=True divides a (S, H, G) tensor by a (G, H, 1) one, as the reference does: a shape error unless
    # the sizes happen to broadcast
    mn = RW_GNN(2, 2, 16, 5, 4, 32, True, 2, 0.0, dev).to(dev)
    with pytest.raises(RuntimeError):
        mn.rw_features(data)


def test_rw_gnn_graph_replay_trains_like_eager(dev):
    """The training step of RW_GNN captured as a HIP graph (gnn_amd.trainer.Trainer with a forward callable)
    follows the eagerly launched one."""
    from gnn_amd.rw_gnn import RW_GNN
    from gnn_amd.trainer import Trainer
    x, ei, batch = _batch(9, 30, 2)
    y = torch.from_numpy(np.random.default_rng(1).integers(0, 2, size=30)).to(dev)
    data = RW_GNN.Data(torch.from_numpy(x).to(dev), torch.from_numpy(ei).to(dev), torch.from_numpy(batch).to(dev))

    def make():
        torch.manual_seed(4)
        m = RW_GNN(2, 2, 16, 5, 4, 32, False, 2, 0.0, dev).to(dev)
        m.train()
        m(data)                                            # creates bn / fc1 / fc2
        return Trainer(m, (None, None, None, y), 1, forward_fn=lambda: m(data))

    eager, graph = make(), make()
    for pe, pg in zip(eager.model.parameters(), graph.model.parameters()):
        assert torch.equal(pe, pg)
    graph.capture()
    for _ in range(3):
        eager.step_eager()
    for _ in range(5):
        eager.step_eager()
        graph.step_graph()
        assert abs(float(eager.loss.detach()) - float(graph.loss.detach())) < 1e-4 * abs(float(eager.loss.detach()))


@pytest.mark.parametrize("h,s,d,steps", [(16, 5, 4, 2), (12, 3, 3, 4), (16, 10, 4, 2), (9, 4, 2, 1)])
def test_keep_s_kernels_against_one_node_hidden_graphs(h, s, d, steps, dev):
    """The keep-S kernel instances and the formulation as H*S one-node hidden graphs on the ordinary kernels give the
    same features and gradients (to rounding: their streams cut the node range differently)."""
    import gnn_amd.rw_gnn as R
    from gnn_amd import _lib
    from gnn_amd.rwconv import propagation_structure
    assert _lib.load().rwk_fused_keep_s_supported(d, h, s, steps) == 1
    x, ei, batch = _batch(h * s, 31, 1)
    n = x.shape[0]
    st = propagation_structure(torch.from_numpy(ei).to(dev), n, torch.from_numpy(batch).to(dev))
    rng = np.random.default_rng(2)
    xs0 = torch.from_numpy(rng.uniform(0, 1, size=(steps, n, d)).astype(np.float32)).to(dev)
    z0 = torch.from_numpy(rng.uniform(0, 1, size=(steps, h, s, d)).astype(np.float32)).to(dev)
    dk = torch.from_numpy(rng.normal(size=(st.n_graphs, steps, h * s)).astype(np.float32)).to(dev)
    runs = []
    try:
        for flag in (True, False):
            R.KEEP_S_KERNELS = flag
            xs, z = xs0.clone().requires_grad_(True), z0.clone().requires_grad_(True)
            k = R.pair_kernel_features(xs, z, st)
            k.backward(dk)
            runs.append((k.detach().cpu().numpy(), xs.grad.cpu().numpy(), z.grad.cpu().numpy()))
    finally:
        R.KEEP_S_KERNELS = True
    for a, b in zip(*runs):
        assert rel_err(a, b.astype(np.float64)) < 1e-5
