"""TU loader and device-side batch construction against the reference's own batches.

The goldens ``mutag_b0`` / ``mutag_b1_eval`` / ``mutag_b2_norm`` hold the inputs the reference's ``load_data`` +
``generate_batches(batch_size=64, shuffle=False)`` produced from its MUTAG files (``oracle/make_golden.py``); the
same files are committed under ``tests/golden/MUTAG`` as data.  Everything here is integer / copy work: bit-exact.
"""
import os

import numpy as np
import pytest
import torch

from _golden import GoldenCase

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
BATCHES = {0: "mutag_b0", 1: "mutag_b1_eval", 2: "mutag_b2_norm"}


def host_batch(adj, features, y, ids):
    """Plain numpy restatement of the loop body of utils.py:60-83."""
    rows, cols, vals, feats, gi = [], [], [], [], []
    off = 0
    for j, g in enumerate(ids):
        m = adj[g].tocoo()
        order = np.lexsort((m.col, m.row))
        rows.append(m.row[order].astype(np.int64) + off)
        cols.append(m.col[order].astype(np.int64) + off)
        vals.append(m.data[order].astype(np.float32))
        feats.append(np.asarray(features[g], dtype=np.float64).astype(np.float32))
        gi.append(np.full(m.shape[0], j, dtype=np.int64))
        off += m.shape[0]
    return (np.vstack([np.concatenate(rows), np.concatenate(cols)]), np.concatenate(vals), np.vstack(feats),
            np.concatenate(gi), np.asarray([y[g] for g in ids], dtype=np.int64))


def test_load_data_matches_reference_batches():
    from gnn_amd.data import load_data
    adj, features, y = load_data("MUTAG", False, root=ROOT)
    assert len(adj) == 188 and sum(m.shape[0] for m in adj) == 3371
    assert sorted(np.unique(y).tolist()) == [-1, 1]
    y = np.unique(y, return_inverse=True)[1]             # main.py:39-40 (LabelEncoder): sorted labels -> 0..C-1
    for b, name in BATCHES.items():
        c = GoldenCase(name)
        ids = list(range(64 * b, min(64 * b + 64, 188)))
        idx, vals, feats, gi, yy = host_batch(adj, features, y, ids)
        assert np.array_equal(idx, c["coo_indices"])
        assert np.array_equal(vals, c["coo_values"])
        assert np.array_equal(feats, c["features"])
        assert np.array_equal(gi, c["graph_indicator"])
        assert np.array_equal(yy, c["y"])


def test_one_hot_node_labels():
    from gnn_amd.data import load_data
    adj, features, _ = load_data("MUTAG", True, root=ROOT)
    labels = np.loadtxt(os.path.join(ROOT, "MUTAG", "MUTAG_node_labels.txt"), dtype=np.int64)
    x = np.vstack(features)
    assert x.shape == (3371, np.unique(labels).size)
    assert np.array_equal(x.sum(axis=1), np.ones(3371))
    assert np.array_equal(np.unique(labels)[x.argmax(axis=1)], labels)      # column order = sorted labels


@pytest.mark.gpu
@pytest.mark.parametrize("shuffle", [False, True])
def test_generate_batches_on_device(shuffle):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from gnn_amd.data import generate_batches, load_data
    dev = torch.device("cuda:0")
    adj, features, y = load_data("MUTAG", False, root=ROOT)
    y = np.unique(y, return_inverse=True)[1]
    np.random.seed(7)
    a_l, f_l, g_l, y_l = generate_batches(adj, features, y, 64, dev, shuffle=shuffle)
    np.random.seed(7)
    index = np.random.permutation(188) if shuffle else np.arange(188)
    assert len(a_l) == 3
    for b in range(3):
        ids = index[64 * b:min(64 * b + 64, 188)]
        idx, vals, feats, gi, yy = host_batch(adj, features, y, ids)
        assert a_l[b].shape == (gi.size, gi.size) and a_l[b].dtype == torch.float32
        assert np.array_equal(a_l[b]._indices().cpu().numpy(), idx)
        assert np.array_equal(a_l[b]._values().cpu().numpy(), vals)
        assert f_l[b].dtype == torch.float32 and np.array_equal(f_l[b].cpu().numpy(), feats)
        assert g_l[b].dtype == torch.int64 and np.array_equal(g_l[b].cpu().numpy(), gi)
        assert y_l[b].dtype == torch.int64 and np.array_equal(y_l[b].cpu().numpy(), yy)
        if not shuffle:                                  # the reference's own batches
            c = GoldenCase(BATCHES[b])
            assert np.array_equal(idx, c["coo_indices"]) and np.array_equal(feats, c["features"])


@pytest.mark.gpu
def test_batches_feed_the_model():
    """End to end on the build's own stack: files -> device batches -> RW_NN forward/backward."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.nn.functional as F
    from gnn_amd import RW_NN
    from gnn_amd.data import generate_batches, load_data
    dev = torch.device("cuda:0")
    adj, features, y = load_data("MUTAG", False, root=ROOT)
    y = np.unique(y, return_inverse=True)[1]
    a_l, f_l, g_l, y_l = generate_batches(adj, features, y, 64, dev)
    torch.manual_seed(0)
    model = RW_NN(1, 2, 16, 5, 4, 32, False, 2, 0.2, dev).to(dev).train()
    out = model(a_l[0], f_l[0], g_l[0])
    F.cross_entropy(out, y_l[0]).backward()
    assert out.shape == (64, 2) and torch.isfinite(out).all()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())


@pytest.mark.gpu
def test_epoch_graph_trains_like_the_step_loop():
    """The reference's epoch over its MUTAG training batches (main.py:103-106) replayed as one HIP graph
    (gnn_amd.trainer.EpochTrainer) against the same steps launched one by one: per-batch losses, parameters, moments
    and running statistics bit for bit after three epochs, dropout on."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from gnn_amd import RW_NN
    from gnn_amd.data import generate_batches, load_data
    from gnn_amd.trainer import EpochTrainer
    dev = torch.device("cuda:0")
    adj, features, y = load_data("MUTAG", False, root=ROOT)
    y = np.unique(y, return_inverse=True)[1]
    batches = list(zip(*generate_batches(adj, features, y, 64, dev)))
    assert len(batches) == 3

    def trainer():
        torch.manual_seed(0)
        m = RW_NN(1, 2, 16, 5, 4, 32, False, 2, 0.2, dev).to(dev).train()
        m.seed_dropout(99)
        return EpochTrainer(m, batches)

    eager, graph = trainer(), trainer()
    graph.capture_epoch()                 # one eager epoch, then the capture
    eager.epoch_eager()
    for _ in range(3):
        le = [float(v) for v in eager.epoch_eager()]
        lg = [float(v) for v in graph.run_epoch()]
        assert le == lg and all(np.isfinite(le))
    graph.opt.set_lr(5e-3)                # a schedule changes the rate between replays
    eager.opt.set_lr(5e-3)
    assert [float(v) for v in eager.epoch_eager()] == [float(v) for v in graph.run_epoch()]
    for a, b in ((eager.opt.flat_param, graph.opt.flat_param), (eager.opt.exp_avg, graph.opt.exp_avg),
                 (eager.opt.exp_avg_sq, graph.opt.exp_avg_sq), (eager.model.bn.running_mean, graph.model.bn.running_mean),
                 (eager.model.bn.running_var, graph.model.bn.running_var), (eager.model._rng_state, graph.model._rng_state)):
        assert torch.equal(a, b)


@pytest.mark.gpu
def test_evaluation_from_the_data_set_against_the_reference_output():
    """The reference's own evaluation-mode output for its second MUTAG batch (golden ``mutag_b1_eval``: made by the
    reference's ``load_data``, ``generate_batches`` and ``RW_NN`` under ``model.eval()``) against the launch that reads
    the graphs straight from the device-resident data set, no batch assembled: files -> DeviceDataset ->
    ``eval_forward_planned``; norm-wise 1e-5, and every element within 1e-5 |ref| + 1e-6 max|ref|."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from gnn_amd import RW_NN
    from gnn_amd.data import DeviceDataset, load_data
    from gnn_amd.eval import eval_forward_planned
    from _golden import rel_err
    dev = torch.device("cuda:0")
    c = GoldenCase("mutag_b1_eval")
    assert not c.train
    adj, features, y = load_data("MUTAG", False, root=ROOT)
    y = np.unique(y, return_inverse=True)[1]
    ds = DeviceDataset(adj, features, y, dev)
    hp = c.hp
    m = RW_NN(hp["input_dim"], hp["max_step"], hp["hidden_graphs"], hp["size_hidden_graphs"], hp["hidden_dim"],
              hp["penultimate_dim"], bool(hp["normalize"]), hp["n_classes"], hp["dropout"], dev)
    m.load_state_dict(c.torch_state(), strict=True)
    m = m.to(dev).eval()
    plan = ds.plan([np.arange(0, 64), np.arange(64, 128)])
    out = eval_forward_planned(m, ds, plan, 1)
    assert out is not None
    got, ref = out.cpu().numpy().astype(np.float64), c["out"].astype(np.float64)
    assert got.shape == ref.shape
    assert rel_err(got, ref) < 1e-5
    assert np.all(np.abs(got - ref) <= 1e-5 * np.abs(ref) + 1e-6 * np.abs(ref).max())


@pytest.mark.gpu
def test_streamed_training_step_against_the_reference_output(monkeypatch):
    """The reference's own training-mode loss and gradients for its first MUTAG batch (golden ``mutag_b0``) against a
    streamed step: files -> DeviceDataset -> ``StreamTrainer.run_pass`` - the pass as one library call, the batch
    assembled inside the forward's first launch.  Loss to 1e-4; gradients to the bound of
    ``test_full_model_against_reference`` (behind train-mode batch-norm over 64 graphs the reference itself is that far
    from the float64 truth)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import gnn_amd.function as fn
    from gnn_amd import RW_NN
    from gnn_amd.data import DeviceDataset, load_data
    from gnn_amd.trainer import StreamTrainer
    from _golden import rel_err
    from test_gpu_parity import _float64_port_grads
    monkeypatch.setattr(fn, "GRAM_FAST_PATH", False)     # the golden ran without dropout: keep the stream kernels
    dev = torch.device("cuda:0")
    c = GoldenCase("mutag_b0")
    assert c.train
    adj, features, y = load_data("MUTAG", False, root=ROOT)
    y = np.unique(y, return_inverse=True)[1]
    ds = DeviceDataset(adj, features, y, dev)
    hp = c.hp
    m = RW_NN(hp["input_dim"], hp["max_step"], hp["hidden_graphs"], hp["size_hidden_graphs"], hp["hidden_dim"],
              hp["penultimate_dim"], bool(hp["normalize"]), hp["n_classes"], hp["dropout"], dev)
    m.load_state_dict(c.torch_state(), strict=True)
    m = m.to(dev).train()
    ids = np.arange(0, 64)
    n, nnz = ds.batch_shape(ids)
    st = StreamTrainer(m, ds, 64, n, nnz, depth=3)
    losses = st.run_pass([ids])
    torch.cuda.synchronize()
    assert st.raw_steps == 1                              # one library call, no tensors, no assembly launch
    assert abs(float(losses[0]) - float(c["loss"])) < 1e-4 * max(1.0, abs(float(c["loss"])))
    truth = _float64_port_grads(c)
    for name, p in m.named_parameters():
        got = p.grad.cpu().numpy()
        ref_own_err = rel_err(c.grads[name], truth[name])
        tiny = float(np.max(np.abs(got - truth[name]))) < 2e-6
        assert tiny or rel_err(got, truth[name]) < max(5e-3, 10 * ref_own_err), name


@pytest.mark.gpu
def test_epoch_loop_over_a_streamed_data_set():
    """The reference's epoch loop (main.py:101-117) on the MUTAG files without resident batches: training passes and
    validation passes over the same slots, three epochs; the loss falls and the validation outputs are proper
    log-probabilities."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from gnn_amd import RW_NN
    from gnn_amd.data import DeviceDataset, load_data
    from gnn_amd.trainer import StreamTrainer
    dev = torch.device("cuda:0")
    adj, features, y = load_data("MUTAG", False, root=ROOT)
    y = np.unique(y, return_inverse=True)[1]
    ds = DeviceDataset(adj, features, y, dev)
    order = np.random.default_rng(0).permutation(188)
    train_ids = [order[i:i + 32] for i in range(0, 160, 32)]
    val_ids = [order[160:188]]
    shapes = [ds.batch_shape(ids) for ids in train_ids + val_ids]
    torch.manual_seed(0)
    model = RW_NN(ds.n_feat, 2, 16, 5, 4, 32, False, 2, 0.2, dev).to(dev)
    model.seed_dropout(5)
    tr = StreamTrainer(model, ds, 32, max(s[0] for s in shapes), max(s[1] for s in shapes), lr=1e-2)
    first = last = None
    for epoch in range(3):
        model.train()
        losses = tr.run_pass(train_ids)
        model.eval()
        outs = tr.evaluate(val_ids)
        torch.cuda.synchronize()
        mean = float(torch.stack(losses).mean())
        first = mean if first is None else first
        last = mean
        assert len(outs) == 1 and outs[0].shape == (28, 2)
        assert torch.allclose(outs[0].exp().sum(1), torch.ones(28, device=dev), atol=1e-5)
        acc = float((outs[0].argmax(1) == ds.y[torch.as_tensor(val_ids[0], device=dev)]).float().mean())
        assert 0.0 <= acc <= 1.0
    assert tr.raw_steps == 15 and np.isfinite(last) and last < first
