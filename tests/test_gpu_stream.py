"""Batches that change every step: the one-launch batch assembly (``DeviceDataset.batch``) against the generic
structure build, and a streamed training pass (``StreamTrainer``: assembly of batch t+1 on a side stream under step
t, reused buffers, no host wait) against the same batches stepped one by one.  Integer work and replays of the same
arithmetic: everything here is bit-exact."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _structure_arrays(st):
    c = st.csr
    out = {"crow": c.crow, "col": c.col, "val": c.val, "gid": st.gid, "seg_ptr": st.seg_ptr, "counts": st.counts,
           "counts_f32": st.counts_f32}
    if c.col16 is not None:
        out["col16"] = c.col16
    return {k: v.cpu().numpy() for k, v in out.items()}


def _assert_same_structure(a, b):
    assert (a.n, a.nnz, a.n_graphs, a.symmetric) == (b.n, b.nnz, b.n_graphs, b.symmetric)
    assert (a.csr.max_degree, a.csr.row_lanes, a.csr.long_threshold, bool(a.csr.unit_values)) == \
        (b.csr.max_degree, b.csr.row_lanes, b.csr.long_threshold, bool(b.csr.unit_values))
    assert (a.csr.col16 is None) == (b.csr.col16 is None)
    assert a.perm is None and b.perm is None and a.csr_t is a.csr and b.csr_t is b.csr
    xa, xb = _structure_arrays(a), _structure_arrays(b)
    assert xa.keys() == xb.keys()
    for k in xa:
        assert xa[k].dtype == xb[k].dtype and np.array_equal(xa[k], xb[k]), k


@pytest.mark.parametrize("n_feat", [None, 3, 16])
def test_batch_build_equals_generic_structure_build(n_feat):
    """The arrays ``rwk_batch_build`` writes next to the batch are those ``GraphStructure.build`` derives from the
    batch's tensors; through fresh tensors and through a reused slot."""
    dev = _dev()
    from gnn_amd.data import DeviceDataset
    from gnn_amd.structure import GraphStructure, get_structure
    from gnn_amd.synth import random_batch
    whole = random_batch(300, 1, 40, 2.2, seed=5, n_feat=n_feat)
    ds = DeviceDataset.from_batch(whole, dev)
    assert ds.symmetric and ds.unit_values
    rng = np.random.default_rng(0)
    slots = ds.make_slots(1, 64, 64 * 40, int(ds.edge_ptr_h[-1]))
    for trial in range(4):
        ids = rng.permutation(300)[:rng.integers(1, 65)]
        for slot in (None, slots[0]):
            adj, feats, gi, y = ds.batch(ids, slot=slot)
            ref = whole_slices(whole, ids)
            assert np.array_equal(adj._indices().cpu().numpy(), ref[0])
            assert np.array_equal(adj._values().cpu().numpy(), ref[1])
            assert np.array_equal(feats.cpu().numpy(), ref[2])
            assert np.array_equal(gi.cpu().numpy(), ref[3])
            assert np.array_equal(y.cpu().numpy(), ref[4])
            assert ds.batch_shape(ids) == (gi.numel(), adj._values().numel())
            st = get_structure(adj, gi)                      # registered by batch(): no build, no device read
            _assert_same_structure(st, GraphStructure.build(adj, gi))


def whole_slices(whole, ids):
    """The batch made of graphs ``ids`` of a synthetic block-diagonal batch, by numpy."""
    starts = np.concatenate([[0], np.cumsum(whole.sizes)])
    r, c = whole.indices
    rows, cols, vals, feats, gi = [], [], [], [], []
    off = 0
    for j, g in enumerate(ids):
        a, b = int(starts[g]), int(starts[g + 1])
        sel = (r >= a) & (r < b)
        rows.append(r[sel] - a + off)
        cols.append(c[sel] - a + off)
        vals.append(whole.values[sel])
        feats.append(whole.features[a:b])
        gi.append(np.full(b - a, j, dtype=np.int64))
        off += b - a
    return (np.vstack([np.concatenate(rows), np.concatenate(cols)]), np.concatenate(vals), np.vstack(feats),
            np.concatenate(gi), whole.y[np.asarray(ids)])


def test_batches_without_edges_and_single_node_graphs():
    """A data set whose graphs have no edge at all (and one node each): the batch, its structure and a training step."""
    dev = _dev()
    import torch.nn.functional as F
    from gnn_amd import RW_NN
    from gnn_amd.data import DeviceDataset
    from gnn_amd.structure import GraphStructure, get_structure
    ds = DeviceDataset.from_arrays(np.arange(6), np.zeros(6, np.int64), np.zeros(0, np.int32), np.zeros(0, np.float32),
                                   np.ones((5, 1), np.float32), [0, 1, 0, 1, 1], dev)
    adj, feats, gi, y = ds.batch([4, 0, 2])
    assert adj.shape == (3, 3) and adj._values().numel() == 0 and gi.cpu().tolist() == [0, 1, 2]
    assert y.cpu().tolist() == [1, 0, 0]
    _assert_same_structure(get_structure(adj, gi), GraphStructure.build(adj, gi))
    torch.manual_seed(0)
    m = RW_NN(1, 2, 4, 3, 4, 8, False, 2, 0.0, dev).to(dev).train()
    F.cross_entropy(m(adj, feats, gi), y).backward()
    assert all(p.grad is None or bool(torch.isfinite(p.grad).all()) for p in m.parameters())
    m.eval()
    assert bool(torch.isfinite(m(adj, feats, gi)).all())


def test_generic_build_reads_the_device_once(monkeypatch):
    """A batch of foreign tensors (row-major sorted COO, sorted ids: what generate_batches produces) costs one
    device-to-host read."""
    dev = _dev()
    from gnn_amd.structure import GraphStructure
    from gnn_amd.synth import random_batch
    adj, _, gi, _ = random_batch(20, 5, 30, 3.0, seed=2).to_torch(dev)
    reads = []
    real_cpu, real_item = torch.Tensor.cpu, torch.Tensor.item
    monkeypatch.setattr(torch.Tensor, "cpu", lambda t, *a, **k: (reads.append("cpu"), real_cpu(t, *a, **k))[1])
    monkeypatch.setattr(torch.Tensor, "item", lambda t: (reads.append("item"), real_item(t))[1])
    st = GraphStructure.build(adj, gi)
    assert reads == ["cpu"] and st.n_graphs == 20 and st.symmetric


def test_asymmetric_data_set_falls_back_to_the_generic_build():
    dev = _dev()
    from gnn_amd.data import DeviceDataset
    from gnn_amd.structure import get_structure
    # two directed 3-cycles
    node_ptr = [0, 3, 6]
    row_ptr = [0, 1, 2, 3, 4, 5, 6]
    col = [1, 2, 0, 4, 5, 3]
    ds = DeviceDataset.from_arrays(node_ptr, row_ptr, col, np.ones(6, np.float32), np.ones((6, 1), np.float32), [0, 1],
                                   dev)
    assert not ds.symmetric
    adj, _, gi, _ = ds.batch([1, 0])
    st = get_structure(adj, gi)
    assert not st.symmetric and st.csr_t is not st.csr
    assert st.csr_t.col.cpu().tolist() == [2, 0, 1, 5, 3, 4]


@pytest.mark.parametrize("shape", ["register", "matrix_core"])
def test_streamed_pass_equals_batches_stepped_one_by_one(shape):
    dev = _dev()
    from gnn_amd import RW_NN
    from gnn_amd.data import DeviceDataset
    from gnn_amd.structure import clear_structure_cache
    from gnn_amd.synth import random_batch
    from gnn_amd.trainer import StreamTrainer, Trainer
    if shape == "register":
        whole = random_batch(640, 20, 60, 3.0, seed=11)
        hp = dict(input_dim=1, max_step=2, hidden_graphs=16, size_hidden_graphs=5, hidden_dim=4)
    else:
        whole = random_batch(640, 10, 40, 2.2, seed=12, n_feat=16)
        hp = dict(input_dim=16, max_step=2, hidden_graphs=64, size_hidden_graphs=10, hidden_dim=64)
    ds = DeviceDataset.from_batch(whole, dev)
    rng = np.random.default_rng(3)
    order = rng.permutation(640)
    id_lists = [order[i:i + 64] for i in range(0, 640, 64)]
    id_lists[3] = id_lists[3][:17]                       # a ragged batch in the middle

    def model():
        torch.manual_seed(0)
        m = RW_NN(hp["input_dim"], hp["max_step"], hp["hidden_graphs"], hp["size_hidden_graphs"], hp["hidden_dim"],
                  32, False, 2, 0.2, dev).to(dev).train()
        m.seed_dropout(77)
        return m

    shapes = [ds.batch_shape(ids) for ids in id_lists]
    st = StreamTrainer(model(), ds, 64, max(s[0] for s in shapes), max(s[1] for s in shapes), depth=3)
    losses_s = st.run_pass(id_lists)
    losses_s += st.run_pass(id_lists[::-1])              # a second pass reuses every slot again
    torch.cuda.synchronize()

    one = Trainer(model(), ds.batch(id_lists[0]))
    losses_o = []
    for ids in id_lists + id_lists[::-1]:
        b = ds.batch(ids)
        clear_structure_cache()                           # the step derives the structure from the tensors itself
        one.adj, one.feats, one.gi, one.y = b
        one.step_eager()
        torch.cuda.synchronize()
        losses_o.append(one.loss.detach())
    assert [float(v) for v in losses_s] == [float(v) for v in losses_o]
    assert all(np.isfinite(float(v)) for v in losses_s)
    for a, b in ((st.opt.flat_param, one.opt.flat_param), (st.opt.exp_avg, one.opt.exp_avg),
                 (st.opt.exp_avg_sq, one.opt.exp_avg_sq), (st.model.bn.running_mean, one.model.bn.running_mean),
                 (st.model.bn.running_var, one.model.bn.running_var), (st.model._rng_state, one.model._rng_state)):
        assert torch.equal(a, b)


@pytest.mark.parametrize("side_stream", [True, False, None])
def test_pass_as_one_library_call_equals_one_call_per_batch(side_stream, monkeypatch):
    """``rwk_model_train_pass`` (every assembly and step of a pass from one host call) against one
    ``rwk_model_train_step`` call per batch: same losses, parameters, moments, statistics and generator state, bit for
    bit - with a ragged batch, a batch the pass call does not serve in the middle (a graph too large for the
    per-graph forward launch: that batch goes through the tensors, the pass call resumes behind it), and a second pass
    over the reused slots."""
    dev = _dev()
    from gnn_amd import RW_NN, composite
    from gnn_amd.data import DeviceDataset
    from gnn_amd.synth import random_batch
    from gnn_amd.trainer import StreamTrainer
    whole = random_batch(320, 20, 60, 3.0, seed=21)
    n0 = whole.n_nodes
    big = 6000                                            # a ring: symmetric, two entries per row
    ring = np.arange(big)
    rows = np.concatenate([whole.indices[0], np.repeat(ring, 2) + n0])
    nb = np.stack([(ring - 1) % big, (ring + 1) % big], axis=1)
    nb.sort(axis=1)
    cols = np.concatenate([whole.indices[1], nb.reshape(-1) + n0])
    n_all = n0 + big
    row_ptr = np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=n_all))]).astype(np.int64)
    node_ptr = np.concatenate([[0], np.cumsum(np.concatenate([whole.sizes, [big]]))]).astype(np.int64)
    feats = np.concatenate([whole.features, np.ones((big, 1), np.float32)]).astype(np.float32)
    y = np.concatenate([whole.y, [1]]).astype(np.int64)
    ds = DeviceDataset.from_arrays(node_ptr, row_ptr, cols.astype(np.int32), np.ones(cols.size, np.float32), feats, y,
                                   dev)
    assert ds.symmetric
    order = np.random.default_rng(5).permutation(320)
    id_lists = [order[i:i + 64] for i in range(0, 320, 64)]
    id_lists[1] = id_lists[1][:23]                       # ragged
    id_lists[2] = np.concatenate([id_lists[2][:40], [320]])   # holds the ring
    shapes = [ds.batch_shape(ids) for ids in id_lists]

    def run(one_call_pass):
        monkeypatch.setattr(composite, "ONE_CALL_PASS", one_call_pass)
        torch.manual_seed(0)
        m = RW_NN(1, 2, 16, 5, 4, 32, False, 2, 0.2, dev).to(dev).train()
        m.seed_dropout(78)
        st = StreamTrainer(m, ds, 64, max(s[0] for s in shapes), max(s[1] for s in shapes), depth=3,
                           side_stream=side_stream)
        losses = st.run_pass(id_lists)
        losses += st.run_pass(id_lists[::-1])
        st.run_pass(id_lists, keep_losses=False)
        torch.cuda.synchronize()
        assert st.raw_steps == 12                         # all but the ring's batch, three passes
        state = [st.opt.flat_param, st.opt.exp_avg, st.opt.exp_avg_sq, st.model.bn.running_mean,
                 st.model.bn.running_var, st.model._rng_state, st.opt.bucket.flat, st.loss.detach().reshape(1)]
        return [float(v) for v in losses], [t.clone() for t in state], [p.grad.clone() for p in m.parameters()]

    la, sa, ga = run(True)
    lb, sb, gb = run(False)
    assert la == lb and all(np.isfinite(v) for v in la)
    for a, b in zip(sa + ga, sb + gb):
        assert torch.equal(a, b)


@pytest.mark.parametrize("n_feat,steps", [(None, 2), (3, 4), (15, 3)])
def test_forward_launch_that_assembles_the_batch_itself(n_feat, steps):
    """``rwk_x_chain_hidden_forward_gather`` (the forward's first launch reading the graphs from the data set) against
    ``rwk_batch_build`` followed by ``rwk_x_chain_hidden_forward``: the same X rows and hidden-graph outputs, and the
    same feature rows, labels, row pointers, compact columns, graph ids, offsets and sizes - bit for bit; with a ragged
    batch in data-set order and one in a shuffled order, single-node graphs and a graph of more than 512 nodes (rows
    beyond a thread's first two) among them."""
    import ctypes
    dev = _dev()
    from gnn_amd import _lib
    from gnn_amd.data import DeviceDataset
    from gnn_amd.synth import random_batch
    kw = {} if n_feat is None else {"n_feat": n_feat}
    whole = random_batch(96, 1, 90, 3.0, seed=31, **kw)
    big = random_batch(2, 600, 700, 3.5, seed=32, **kw)
    rows = np.concatenate([whole.indices[0], big.indices[0] + whole.n_nodes])
    cols = np.concatenate([whole.indices[1], big.indices[1] + whole.n_nodes]).astype(np.int32)
    n_all = whole.n_nodes + big.n_nodes
    row_ptr = np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=n_all))]).astype(np.int64)
    node_ptr = np.concatenate([[0], np.cumsum(np.concatenate([whole.sizes, big.sizes]))]).astype(np.int64)
    feats = np.concatenate([whole.features, big.features]).astype(np.float32)
    y = np.concatenate([whole.y, big.y]).astype(np.int64)
    ds = DeviceDataset.from_arrays(node_ptr, row_ptr, cols, np.ones(cols.size, np.float32), feats, y, dev)
    assert ds.symmetric and ds.unit_values
    f = ds.n_feat
    lib = _lib.load()
    stream = _lib.current_stream(dev)
    h, s = 6, 5
    torch.manual_seed(3)
    w = torch.randn(4, f, device=dev)
    b = torch.randn(4, device=dev)
    adj_hidden = torch.randn(h, s * (s - 1) // 2, device=dev)
    feat_hidden = torch.randn(h, s, 4, device=dev)

    class Gather(ctypes.Structure):
        _fields_ = [(k, ctypes.c_void_p) for k in ("table", "ds_row_ptr", "ds_col", "ds_feat", "ds_y", "feat", "y",
                                                   "crow", "col16", "gid", "seg_ptr", "counts_f32")]

    order = np.random.default_rng(7).permutation(98)
    for ids in (np.arange(5, 70), order[:41], np.array([97, 3, 96, 11])):
        plan = ds.plan([ids])
        adj, fe, gi, yy = ds.batch_planned(plan, 0)
        from gnn_amd.structure import get_structure
        st = get_structure(adj, gi)
        bsz, n, nnz, _, fits16, off, _, _, max_graph = plan.metas[0]
        assert fits16 and st.csr.col16 is not None
        xs_a = torch.empty(steps, n, 4, device=dev)
        ah_a, z_a = torch.empty(h, s, s, device=dev), torch.empty(steps, h, s, 4, device=dev)
        _lib.check(lib.rwk_x_chain_hidden_forward(
            fe.data_ptr(), w.data_ptr(), b.data_ptr(), n, f, st.csr.crow.data_ptr(), st.csr.col.data_ptr(), None,
            st.seg_ptr.data_ptr(), bsz, max_graph, steps, xs_a.data_ptr(), adj_hidden.data_ptr(),
            feat_hidden.data_ptr(), h, s, ah_a.data_ptr(), z_a.data_ptr(), None, None, stream), "x_chain")
        out = {"feat": torch.full((n, f), -1.0, device=dev), "y": torch.full((bsz,), -1, dtype=torch.int64, device=dev),
               "crow": torch.full((n + 1,), -1, dtype=torch.int32, device=dev),
               "col16": torch.full((nnz,), -1, dtype=torch.int16, device=dev),
               "gid": torch.full((n,), -1, dtype=torch.int32, device=dev),
               "seg_ptr": torch.full((bsz + 1,), -1, dtype=torch.int32, device=dev),
               "counts_f32": torch.full((bsz,), -1.0, device=dev)}
        g = Gather()
        g.table = plan.tables.data_ptr() + 8 * off
        g.ds_row_ptr, g.ds_col = ds.row_ptr.data_ptr(), ds.col.data_ptr()
        g.ds_feat, g.ds_y = ds.features.data_ptr(), ds.y.data_ptr()
        for k, t in out.items():
            setattr(g, k, t.data_ptr())
        xs_b = torch.empty(steps, n, 4, device=dev)
        ah_b, z_b = torch.empty(h, s, s, device=dev), torch.empty(steps, h, s, 4, device=dev)
        _lib.check(lib.rwk_x_chain_hidden_forward_gather(
            ctypes.byref(g), w.data_ptr(), b.data_ptr(), n, f, bsz, max_graph, steps, xs_b.data_ptr(),
            adj_hidden.data_ptr(), feat_hidden.data_ptr(), h, s, ah_b.data_ptr(), z_b.data_ptr(), None, None, stream),
            "x_chain gather")
        torch.cuda.synchronize()
        assert torch.equal(xs_a, xs_b) and torch.equal(ah_a, ah_b) and torch.equal(z_a, z_b)
        want = {"feat": fe, "y": yy, "crow": st.csr.crow, "col16": st.csr.col16, "gid": st.gid, "seg_ptr": st.seg_ptr,
                "counts_f32": st.counts_f32}
        for k in out:
            assert torch.equal(out[k], want[k].reshape(out[k].shape)), k


@pytest.mark.parametrize("n_feat,steps,normalize", [(None, 2, False), (3, 4, True), (15, 1, False)])
def test_evaluation_forward_straight_from_the_data_set(n_feat, steps, normalize):
    """``eval_forward_planned`` (one launch that reads its graphs from the data set, no batch assembled) against the
    model called in evaluation mode on the assembled batch: the same log-probabilities, bit for bit."""
    dev = _dev()
    from gnn_amd import RW_NN
    from gnn_amd.data import DeviceDataset
    from gnn_amd.eval import eval_forward_planned
    from gnn_amd.synth import random_batch
    kw = {} if n_feat is None else {"n_feat": n_feat}
    whole = random_batch(96, 1, 90, 3.0, seed=41, **kw)
    big = random_batch(2, 600, 700, 3.5, seed=42, **kw)
    rows = np.concatenate([whole.indices[0], big.indices[0] + whole.n_nodes])
    cols = np.concatenate([whole.indices[1], big.indices[1] + whole.n_nodes]).astype(np.int32)
    n_all = whole.n_nodes + big.n_nodes
    row_ptr = np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=n_all))]).astype(np.int64)
    node_ptr = np.concatenate([[0], np.cumsum(np.concatenate([whole.sizes, big.sizes]))]).astype(np.int64)
    feats = np.concatenate([whole.features, big.features]).astype(np.float32)
    y = np.concatenate([whole.y, big.y]).astype(np.int64)
    ds = DeviceDataset.from_arrays(node_ptr, row_ptr, cols, np.ones(cols.size, np.float32), feats, y, dev)
    torch.manual_seed(5)
    m = RW_NN(ds.n_feat, steps, 16, 5, 4, 32, normalize, 3, 0.2, dev).to(dev)
    with torch.no_grad():
        m.bn.running_mean.uniform_(-0.5, 0.5)
        m.bn.running_var.uniform_(0.5, 2.0)
    order = np.random.default_rng(9).permutation(98)
    id_lists = [np.arange(5, 70), order[:41], np.array([97, 3, 96, 11])]
    plan = ds.plan(id_lists)
    m.train()
    assert eval_forward_planned(m, ds, plan, 0) is None         # training mode: not this launch's business
    m.eval()
    for i in range(len(id_lists)):
        adj, fe, gi, _ = ds.batch_planned(plan, i)
        with torch.no_grad():
            want = m(adj, fe, gi)
        got = eval_forward_planned(m, ds, plan, i)
        assert got is not None and got.shape == want.shape
        torch.cuda.synchronize()
        assert torch.equal(got, want)
        assert np.isfinite(got.cpu().numpy()).all()


def test_stream_trainer_evaluates_batches_of_the_data_set():
    """``StreamTrainer.evaluate``: outputs equal to the model called on the assembled batches, in evaluation mode
    (one launch per batch, nothing assembled) and in training mode with dropout off (assembled into the slots), and a
    training pass still runs after it on the same slots."""
    dev = _dev()
    from gnn_amd import RW_NN
    from gnn_amd.data import DeviceDataset
    from gnn_amd.synth import random_batch
    from gnn_amd.trainer import StreamTrainer
    whole = random_batch(200, 10, 50, 3.0, seed=51)
    ds = DeviceDataset.from_batch(whole, dev)
    id_lists = [np.arange(i, i + 40) for i in range(0, 200, 40)]
    shapes = [ds.batch_shape(ids) for ids in id_lists]
    torch.manual_seed(0)
    m = RW_NN(1, 2, 16, 5, 4, 32, False, 2, 0.0, dev).to(dev).train()
    st = StreamTrainer(m, ds, 40, max(s[0] for s in shapes), max(s[1] for s in shapes), depth=3)
    st.run_pass(id_lists)
    for mode in ("eval", "train"):
        getattr(m, mode)()
        if mode == "train":
            running = [m.bn.running_mean.clone(), m.bn.running_var.clone(), m.bn.num_batches_tracked.clone()]
        outs = st.evaluate(id_lists)
        if mode == "train":       # the reference calls with a fresh batch norm state each time; compare like with like
            m.bn.running_mean.copy_(running[0]); m.bn.running_var.copy_(running[1])
            m.bn.num_batches_tracked.copy_(running[2])
        assert len(outs) == len(id_lists)
        for ids, out in zip(id_lists, outs):
            adj, fe, gi, _ = ds.batch(ids)
            with torch.no_grad():
                want = m(adj, fe, gi)
            torch.cuda.synchronize()
            assert out.shape == want.shape and not out.requires_grad
            assert torch.equal(out, want), mode
            if mode == "train":
                break             # (each training-mode call moves the running statistics: one batch is enough)
    m.train()
    losses = st.run_pass(id_lists)
    torch.cuda.synchronize()
    assert all(np.isfinite(float(v)) for v in losses)
