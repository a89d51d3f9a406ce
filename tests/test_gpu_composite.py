"""The training step of the eager loop as one autograd node with one library call per direction
(``rwk_model_forward`` / ``rwk_model_backward``) against the separate nodes of the layer and the head: the same
launches in the same order, so outputs, gradients, running statistics and generator state are compared bit for bit."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

#        graphs, nodes, n_feat, H, S, D, L, normalize, dropout, weighted
CASES = [(64, (20, 60), None, 16, 5, 4, 2, False, 0.2, False),
         (40, (5, 90), 3, 16, 5, 4, 3, True, 0.3, False),
         (33, (2, 40), None, 8, 3, 4, 4, False, 0.5, True),
         (64, (20, 60), 15, 64, 4, 4, 2, False, 0.2, False),
         (20, (10, 30), None, 5, 1, 4, 2, True, 0.1, False),
         (64, (20, 60), None, 16, 5, 4, 2, False, 0.0, False)]


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _batch(dev, graphs, nodes, n_feat, weighted, seed):
    from gnn_amd.synth import random_batch
    b = random_batch(graphs, nodes[0], nodes[1], 2.5, seed=seed, n_feat=n_feat)
    adj, feats, gi, y = b.to_torch(dev)
    if weighted:                                   # symmetric weights: w(r, c) = w(c, r)
        adj = adj.coalesce()
        r, c = adj.indices()
        w = 0.5 + ((torch.minimum(r, c) * 31 + torch.maximum(r, c) * 17) % 13).float() / 13.0
        adj = torch.sparse_coo_tensor(adj.indices(), w, adj.shape).coalesce()
    return adj, feats, gi, y


def _run(dev, case, composite, monkeypatch, steps=3):
    import gnn_amd
    import gnn_amd.composite as comp
    import gnn_amd.function as fn
    from gnn_amd import RW_NN
    from gnn_amd.structure import clear_structure_cache
    graphs, nodes, n_feat, h, s, d, steps_l, normalize, p, weighted = case
    monkeypatch.setattr(comp, "COMPOSITE_STEP", composite)
    if p == 0.0:
        monkeypatch.setattr(fn, "GRAM_FAST_PATH", False)
    clear_structure_cache()
    adj, feats, gi, y = _batch(dev, graphs, nodes, n_feat, weighted, seed=7)
    torch.manual_seed(1)
    m = RW_NN(feats.shape[1], steps_l, h, s, d, 32, normalize, 3, p, dev).to(dev).train()
    m.seed_dropout(99)
    opt = gnn_amd.Adam(m.parameters(), lr=1e-2)
    calls = []
    real = comp._RWModelStep.apply
    monkeypatch.setattr(comp._RWModelStep, "apply", lambda *a: (calls.append(1), real(*a))[1])
    outs, grads = [], []
    for _ in range(steps):
        opt.zero_grad()
        out = m(adj, feats, gi)
        loss = F.cross_entropy(out, y % 3)
        loss.backward()
        outs.append(out.detach().clone())
        grads.append([None if q.grad is None else q.grad.detach().clone() for q in m.parameters()])
        opt.step()
    torch.cuda.synchronize()
    state = [q.detach().clone() for q in m.parameters()] + [m.bn.running_mean.clone(), m.bn.running_var.clone(),
                                                            m.bn.num_batches_tracked.clone(), m._rng_state.clone()]
    return outs, grads, state, len(calls)


@pytest.mark.parametrize("case", CASES, ids=lambda c: "g%d_f%s_h%d_s%d_l%d_p%s%s" % (
    c[0], c[2], c[3], c[4], c[6], c[8], "_w" if c[9] else ""))
def test_composite_step_equals_separate_nodes(case, monkeypatch):
    dev = _dev()
    o1, g1, s1, n1 = _run(dev, case, True, monkeypatch)
    o0, g0, s0, n0 = _run(dev, case, False, monkeypatch)
    assert n1 == 3 and n0 == 0                     # the composite node ran, and only when asked for
    for a, b in zip(o1, o0):
        assert torch.equal(a, b) and bool(torch.isfinite(a).all())
    for ga, gb in zip(g1, g0):
        for a, b in zip(ga, gb):
            assert (a is None) == (b is None)
            if a is not None:
                assert a.shape == b.shape and torch.equal(a, b)
    for a, b in zip(s1, s0):
        assert torch.equal(a, b)


def test_composite_step_declines_what_it_does_not_serve(monkeypatch):
    """Unsorted graph ids, wide inputs, hidden_dim 8, an armed step tail and evaluation mode stay on the separate
    nodes; the results are still right (finite, same shape)."""
    dev = _dev()
    import gnn_amd.composite as comp
    from gnn_amd import RW_NN
    calls = []
    real = comp._RWModelStep.apply
    monkeypatch.setattr(comp._RWModelStep, "apply", lambda *a: (calls.append(1), real(*a))[1])
    adj, feats, gi, y = _batch(dev, 32, (10, 40), None, False, seed=3)
    torch.manual_seed(0)
    wide = RW_NN(1, 2, 16, 5, 8, 32, False, 2, 0.2, dev).to(dev).train()
    assert wide(adj, feats, gi).shape == (32, 2) and not calls
    m = RW_NN(1, 2, 16, 5, 4, 32, False, 2, 0.2, dev).to(dev).train()
    perm = torch.randperm(gi.numel(), device=dev)
    inv = torch.argsort(perm)
    a = adj.coalesce()
    adj_p = torch.sparse_coo_tensor(torch.stack([inv[a.indices()[0]], inv[a.indices()[1]]]), a.values(), a.shape)
    assert m(adj_p.coalesce(), feats[perm], gi[perm]).shape == (32, 2) and not calls
    m.eval()
    assert m(adj, feats, gi).shape == (32, 2) and not calls
    m.train()
    feats_g = feats.clone().requires_grad_(True)
    m(adj, feats_g, gi).sum().backward()
    assert feats_g.grad is not None and not calls
    assert m(adj, feats, gi).shape == (32, 2) and calls == [1]


TRAIN_CASES = [(64, (20, 60), None, 16, 5, 4, 2, False, 0.2, False),
               (40, (5, 90), 3, 16, 5, 4, 3, True, 0.3, False),
               (33, (2, 40), None, 8, 3, 4, 4, False, 0.5, True),
               (64, (20, 60), None, 16, 5, 4, 2, False, 0.0, False)]


def _train(dev, case, one_call, monkeypatch, world=1, graph=False):
    import gnn_amd.composite as comp
    import gnn_amd.function as fn
    from gnn_amd import RW_NN
    from gnn_amd.structure import clear_structure_cache
    from gnn_amd.trainer import Trainer
    graphs, nodes, n_feat, h, s, d, steps_l, normalize, p, weighted = case
    monkeypatch.setattr(comp, "ONE_CALL_TRAIN_STEP", one_call)
    if p == 0.0:
        monkeypatch.setattr(fn, "GRAM_FAST_PATH", False)
    clear_structure_cache()
    adj, feats, gi, y = _batch(dev, graphs, nodes, n_feat, weighted, seed=7)
    torch.manual_seed(1)
    m = RW_NN(feats.shape[1], steps_l, h, s, d, 32, normalize, 3, p, dev).to(dev).train()
    m.seed_dropout(99)
    tr = Trainer(m, (adj, feats, gi, y % 3), world)
    calls = []
    real = comp.TrainStep.run
    monkeypatch.setattr(comp.TrainStep, "run", lambda self, *a: (calls.append(1), real(self, *a))[1])
    losses, grads = [], []
    if graph:
        tr.capture(steps_per_graph=2)
        for _ in range(2):
            tr.step_graph()
            losses.append(tr.loss.detach().clone())
        tr.run_steps(4)
        losses += [v.detach().clone() for v in tr.losses]
    else:
        for _ in range(4):
            tr.step_eager()
            losses.append(tr.loss.detach().clone())
            grads.append([q.grad.detach().clone() for q in m.parameters()])
    torch.cuda.synchronize()
    state = [q.detach().clone() for q in m.parameters()] + [
        tr.opt.exp_avg.clone(), tr.opt.exp_avg_sq.clone(), tr.bucket.flat.clone(), m.bn.running_mean.clone(),
        m.bn.running_var.clone(), m.bn.num_batches_tracked.clone(), m._rng_state.clone()]
    return losses, grads, state, len(calls)


@pytest.mark.parametrize("case", TRAIN_CASES, ids=lambda c: "g%d_f%s_h%d_s%d_l%d_p%s%s" % (
    c[0], c[2], c[3], c[4], c[6], c[8], "_w" if c[9] else ""))
def test_one_call_train_step_equals_the_autograd_step(case, monkeypatch):
    """``rwk_model_train_step`` against ``Trainer``'s step through autograd: losses, per-parameter gradients, flat
    gradient, parameters, moments, running statistics and generator state after four eager steps."""
    dev = _dev()
    l1, g1, s1, n1 = _train(dev, case, True, monkeypatch)
    l0, g0, s0, n0 = _train(dev, case, False, monkeypatch)
    assert n1 >= 4 and n0 == 0
    assert [float(v) for v in l1] == [float(v) for v in l0] and all(np.isfinite(float(v)) for v in l1)
    for ga, gb in zip(g1, g0):
        for a, b in zip(ga, gb):
            assert a.shape == b.shape and torch.equal(a, b)
    for a, b in zip(s1, s0):
        assert torch.equal(a, b)


def test_one_call_train_step_under_capture_and_in_the_data_parallel_shape(monkeypatch):
    """Captured (single and multi-step replays) and with the update left to a separate launch (the shape of a
    data-parallel step, here with one process): same bits as the autograd step."""
    dev = _dev()
    case = TRAIN_CASES[0]
    for kw in (dict(graph=True), dict(world=2)):
        l1, _, s1, n1 = _train(dev, case, True, monkeypatch, **kw)
        l0, _, s0, n0 = _train(dev, case, False, monkeypatch, **kw)
        assert n1 > 0 and n0 == 0
        assert [float(v) for v in l1] == [float(v) for v in l0]
        for a, b in zip(s1, s0):
            assert torch.equal(a, b)
