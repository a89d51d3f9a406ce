"""The layer without dropout through per-graph Gram matrices (csrc/rwk_gram.hip) against the stream kernels with the
identity mask and against the float64 oracle: unsorted graph ids, ragged widths (hidden_dim 1..4), one to four steps,
normalisation, single-node graphs; bitwise reproducibility; the training-mode model with dropout 0 and the evaluation
forward of rwgnn2/main.py:110-117."""
import numpy as np
import pytest
import torch

from _golden import rel_err
from oracle import rwnn_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _case(seed, n_graphs, n_lo, n_hi, unsorted):
    from gnn_amd.synth import random_batch
    b = random_batch(n_graphs, n_lo, n_hi, 2.5, seed=seed)
    gi = b.graph_indicator.copy()
    idx = b.indices.copy()
    if unsorted:                      # a random relabelling of the nodes: graph ids no longer sorted
        perm = np.random.default_rng(seed).permutation(b.n_nodes)
        inv = np.empty_like(perm)
        inv[perm] = np.arange(b.n_nodes)
        gi = gi[perm]
        r, c = inv[idx[0]], inv[idx[1]]
        order = np.lexsort((c, r))
        idx = np.vstack([r[order], c[order]])
    return b.n_nodes, gi, idx, b.values


@pytest.mark.parametrize("h,s,d,steps,normalize,unsorted", [
    (16, 5, 4, 2, False, False), (16, 5, 4, 2, True, True), (7, 3, 1, 1, False, False), (33, 12, 3, 4, True, False),
    (64, 2, 2, 3, False, True), (1, 1, 4, 2, True, False), (20, 6, 4, 4, False, True)])
def test_gram_path_against_oracle_and_stream_kernels(h, s, d, steps, normalize, unsorted, dev, monkeypatch):
    import gnn_amd.function as fn
    from gnn_amd import get_structure, rw_kernel_layer
    n, gi, idx, val = _case(h * 7 + s, 41, 1, 70, unsorted)
    rng = np.random.default_rng(h + 100 * d + steps)
    pre = rng.normal(size=(n, d)).astype(np.float32)
    adj_h = rng.uniform(-1, 1, size=(h, s * (s - 1) // 2)).astype(np.float32)
    z0 = rng.uniform(0, 1, size=(h, s, d)).astype(np.float32)
    g = int(gi.max()) + 1
    dk = rng.normal(size=(g, steps * h)).astype(np.float32)
    adj = torch.sparse_coo_tensor(torch.from_numpy(idx), torch.from_numpy(val), (n, n)).to(dev)
    st = get_structure(adj, torch.from_numpy(gi).to(dev))
    assert (st.perm is not None) == unsorted

    def run(gram):
        monkeypatch.setattr(fn, "GRAM_FAST_PATH", gram)
        t = [torch.from_numpy(a).to(dev).requires_grad_(True) for a in (pre, adj_h, z0)]
        k = rw_kernel_layer(t[0], t[1], t[2], st, steps, normalize, 0.0, None)
        k.backward(torch.from_numpy(dk).to(dev))
        return [k.detach()] + [None if x.grad is None else x.grad for x in t]

    got, old = run(True), run(False)
    again = run(True)
    csr = O.coo_to_csr(idx, val, n)
    x0 = 1.0 / (1.0 + np.exp(-pre.astype(np.float64)))
    k_ref = O.rw_kernel_forward(csr, x0, adj_h, z0, gi, steps, normalize)
    d_x0, d_adj, d_feat = O.rw_kernel_backward(csr, x0, adj_h, z0, gi, steps, dk, normalize)
    want = [k_ref, d_x0 * x0 * (1 - x0), d_adj if (steps > 1 and s > 1) else None, d_feat]
    for a, b, c, w in zip(got, old, again, want):
        if w is None:
            continue
        assert rel_err(a.cpu().numpy(), w) < TOL
        assert rel_err(b.cpu().numpy(), w) < TOL
        assert torch.equal(a, c), "the Gram path must be bitwise reproducible"


def test_model_without_dropout_trains_and_evaluates_through_the_gram_path(dev, monkeypatch):
    """RW_NN in training mode with dropout 0 (loss and every gradient) and in evaluation mode (scores), Gram path
    against stream kernels."""
    import torch.nn.functional as F
    import gnn_amd.function as fn
    from gnn_amd import RW_NN
    from gnn_amd.synth import random_batch
    adj, feats, gi, y = random_batch(50, 5, 60, 3.0, seed=8).to_torch(dev)

    def run(gram, train):
        monkeypatch.setattr(fn, "GRAM_FAST_PATH", gram)
        torch.manual_seed(3)
        m = RW_NN(1, 2, 16, 5, 4, 32, False, 2, 0.0, dev).to(dev)
        m.train(train)
        if not train:
            with torch.no_grad():
                return [m(adj, feats, gi)]
        out = m(adj, feats, gi)
        loss = F.cross_entropy(out, y)
        loss.backward()
        return [out.detach(), loss.detach()] + [p.grad for p in m.parameters()]

    for train in (True, False):
        a, b = run(True, train), run(False, train)
        for u, v in zip(a, b):
            assert rel_err(u.cpu().numpy(), v.cpu().numpy()) < 2e-5


def test_training_step_without_dropout_replays_like_eager(dev):
    """Trainer on the Gram path (dropout 0): the captured step replayed against the eager step - loss, parameters,
    moments bit for bit."""
    import bench
    from gnn_amd import RW_NN
    from gnn_amd.synth import random_batch
    batch = random_batch(64, 10, 50, 3.0, seed=4).to_torch(dev)

    def trainer():
        torch.manual_seed(2)
        m = RW_NN(1, 2, 16, 5, 4, 32, False, 2, 0.0, dev).to(dev).train()
        return bench.Trainer(m, batch)

    eager, graph = trainer(), trainer()
    graph.capture()
    for _ in range(3):
        eager.step_eager()
    for _ in range(4):
        eager.step_eager()
        graph.step_graph()
    torch.cuda.synchronize()
    assert float(eager.loss) == float(graph.loss) and np.isfinite(float(eager.loss))
    for a, b in ((eager.opt.flat_param, graph.opt.flat_param), (eager.opt.exp_avg, graph.opt.exp_avg),
                 (eager.opt.exp_avg_sq, graph.opt.exp_avg_sq)):
        assert torch.equal(a, b)
