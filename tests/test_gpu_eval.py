"""The evaluation-mode forward as one launch (csrc/rwk_eval.hip, gnn_amd/eval.py) against the separate launches (which
the golden cases pin to the reference): scores to 1e-5, over input widths on both sides of the projection kernel's
limit, one to four steps, ragged hidden_dim, normalisation, single-node graphs; gradients through its recomputing
backward; bitwise reproducibility; and the inputs it must leave to the separate launches (unsorted graph ids, an entry
that joins two graphs, a graph too large for the LDS)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from _golden import GoldenCase, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _model(dev, f, steps, h, s, d, normalize, nc=3, p=32):
    from gnn_amd import RW_NN
    torch.manual_seed(5)
    m = RW_NN(f, steps, h, s, d, p, normalize, nc, 0.2, dev).to(dev)
    with torch.no_grad():                                  # running statistics that are not the initial 0 / 1
        m.bn.running_mean.uniform_(-0.5, 0.5)
        m.bn.running_var.uniform_(0.5, 2.0)
        m.bn.weight.uniform_(0.5, 1.5)
        m.bn.bias.uniform_(-0.2, 0.2)
    return m.eval()


def _both(m, adj, feats, gi, monkeypatch, grad=False):
    import gnn_amd.eval as ev
    outs = []
    for fused in (True, False):
        monkeypatch.setattr(ev, "FUSED_EVAL_FORWARD", fused)
        outs.append(m(adj, feats, gi) if grad else m(adj, feats, gi).detach())
    return outs


@pytest.mark.parametrize("f,steps,h,s,d,normalize", [
    (1, 2, 16, 5, 4, False), (1, 2, 16, 5, 4, True), (3, 1, 7, 3, 2, False), (16, 4, 33, 6, 3, True),
    (20, 3, 8, 4, 4, False), (2, 2, 64, 12, 1, False), (1, 3, 5, 1, 4, True)])
def test_one_launch_forward_equals_separate_launches(f, steps, h, s, d, normalize, dev, monkeypatch):
    from gnn_amd.eval import eval_forward_supported
    from gnn_amd.structure import get_structure
    from gnn_amd.synth import random_batch
    adj, feats, gi, _ = random_batch(37, 1, 90, 2.6, seed=f + 10 * steps, n_feat=None if f == 1 else f).to_torch(dev)
    m = _model(dev, f, steps, h, s, d, normalize)
    assert eval_forward_supported(m, get_structure(adj, gi), feats)
    fused, separate = _both(m, adj, feats, gi, monkeypatch)
    assert fused.shape == separate.shape == (37, 3)
    assert rel_err(fused.cpu().numpy(), separate.cpu().numpy()) < 1e-5
    assert abs(float(torch.logsumexp(fused, 1).abs().max())) < 1e-5       # rows are log-probabilities
    again, _ = _both(m, adj, feats, gi, monkeypatch)
    assert torch.equal(fused, again)


def test_reference_evaluation_case(dev):
    """The reference's own evaluation-mode output (golden mutag_b1_eval) through the one launch."""
    from gnn_amd import RW_NN
    from gnn_amd.eval import eval_forward_supported
    from gnn_amd.structure import get_structure
    c = GoldenCase("mutag_b1_eval")
    assert not c.train
    hp = c.hp
    m = RW_NN(hp["input_dim"], hp["max_step"], hp["hidden_graphs"], hp["size_hidden_graphs"], hp["hidden_dim"],
              hp["penultimate_dim"], bool(hp["normalize"]), hp["n_classes"], hp["dropout"], dev)
    m.load_state_dict(c.torch_state())
    m = m.to(dev).eval()
    adj, feats, gi, _ = c.torch_inputs(dev)
    assert eval_forward_supported(m, get_structure(adj, gi), feats)
    with torch.no_grad():
        out = m(adj, feats, gi)
    assert rel_err(out.cpu().numpy(), c["out"]) < 1e-5


def test_backward_through_the_one_launch_forward(dev, monkeypatch):
    """main.py:110-117 records a graph under model.eval(); a backward through it gives the separate launches'
    gradients."""
    from gnn_amd.synth import random_batch
    adj, feats, gi, y = random_batch(20, 3, 40, 3.0, seed=3, n_classes=3).to_torch(dev)
    m = _model(dev, 1, 2, 16, 5, 4, False)
    grads = []
    for out in _both(m, adj, feats, gi, monkeypatch, grad=True):
        m.zero_grad()
        F.nll_loss(out, y).backward()
        grads.append([p.grad.clone() for p in m.parameters()])
    for a, b in zip(*grads):
        assert rel_err(a.cpu().numpy(), b.cpu().numpy()) < 1e-5


def test_inputs_left_to_the_separate_launches(dev, monkeypatch):
    from gnn_amd.eval import eval_forward_supported
    from gnn_amd.structure import get_structure
    from gnn_amd.synth import random_batch
    b = random_batch(9, 4, 30, 2.5, seed=6)
    m = _model(dev, 1, 2, 16, 5, 4, False)
    # unsorted graph ids
    perm = np.random.default_rng(1).permutation(b.n_nodes)
    inv = np.empty_like(perm)
    inv[perm] = np.arange(b.n_nodes)
    r, c = inv[b.indices[0]], inv[b.indices[1]]
    order = np.lexsort((c, r))
    adj = torch.sparse_coo_tensor(torch.from_numpy(np.vstack([r[order], c[order]])), torch.from_numpy(b.values),
                                  (b.n_nodes, b.n_nodes)).to(dev)
    gi = torch.from_numpy(b.graph_indicator[perm]).to(dev)
    feats = torch.from_numpy(b.features[perm]).to(dev)
    assert not eval_forward_supported(m, get_structure(adj, gi), feats)
    out_perm = m(adj, feats, gi).detach()
    adj0, feats0, gi0, _ = b.to_torch(dev)
    assert rel_err(out_perm.cpu().numpy(), m(adj0, feats0, gi0).detach().cpu().numpy()) < 1e-5
    # one stored entry joins graph 0 and graph 8
    idx = np.hstack([b.indices, np.array([[0, b.n_nodes - 1], [b.n_nodes - 1, 0]])])
    order = np.lexsort((idx[1], idx[0]))
    adj_x = torch.sparse_coo_tensor(torch.from_numpy(idx[:, order]), torch.ones(idx.shape[1]),
                                    (b.n_nodes, b.n_nodes)).to(dev)
    assert not eval_forward_supported(m, get_structure(adj_x, gi0), feats0)
    fused, separate = _both(m, adj_x, feats0, gi0, monkeypatch)
    assert torch.equal(fused, separate)                    # both ran the separate launches
    # a graph of 9000 nodes with 64 x 12 hidden nodes does not fit the LDS
    big = random_batch(2, 9000, 9000, 2.5, seed=7)
    adj_b, feats_b, gi_b, _ = big.to_torch(dev)
    m_big = _model(dev, 1, 4, 64, 12, 4, False)
    assert not eval_forward_supported(m_big, get_structure(adj_b, gi_b), feats_b)
    assert torch.isfinite(m_big(adj_b, feats_b, gi_b)).all()
