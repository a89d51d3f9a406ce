"""GPU tests of the helper kernels around the layer: deterministic row reduction, fused input projection,
flat Adam, HIP-graph replay of a whole training step."""
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


@pytest.fixture(autouse=True)
def stream_kernels_without_dropout(monkeypatch):
    """This file exercises the launch structure of the training step on the stream kernels (statistics hand-over,
    folded head finish, optimiser in the last launch) with and without dropout; without dropout the layer would
    otherwise take the Gram-matrix path, which has its own tests (test_gpu_gram.py).  The step goes through autograd
    here: the one-call step (composite.TrainStep), which replaces it where it applies, is compared with this path
    in test_gpu_composite.py."""
    import gnn_amd.composite as comp
    import gnn_amd.function as fn
    monkeypatch.setattr(fn, "GRAM_FAST_PATH", False)
    monkeypatch.setattr(comp, "ONE_CALL_TRAIN_STEP", False)


@pytest.mark.parametrize("p,v", [(1, 7), (32, 640), (33, 5), (1203, 640), (5000, 70)])
def test_reduce_rows(p, v, dev):
    from gnn_amd import _lib
    lib = _lib.load()
    x = torch.randn(p, v, device=dev)
    out = torch.empty(v, device=dev)
    ws = torch.empty(max(1, lib.rwk_reduce_rows_workspace(p, v)), device=dev)
    stream = _lib.current_stream(dev)
    _lib.check(lib.rwk_reduce_rows(x.data_ptr(), p, v, out.data_ptr(), ws.data_ptr(), stream), "reduce")
    want = x.double().sum(0).cpu().numpy()
    assert rel_err(out.cpu().numpy(), want) < 1e-5
    out2 = torch.empty(v, device=dev)
    _lib.check(lib.rwk_reduce_rows(x.data_ptr(), p, v, out2.data_ptr(), ws.data_ptr(), stream), "reduce")
    assert torch.equal(out, out2)


@pytest.mark.parametrize("n,f,d", [(1000, 1, 4), (777, 3, 6), (64, 8, 16), (5000, 2, 64), (3000, 16, 64),
                                   (901, 11, 4), (2500, 16, 4)])
def test_projection_forward_backward(n, f, d, dev):
    from gnn_amd import _lib
    lib = _lib.load()
    g = torch.Generator(device="cpu").manual_seed(n + f + d)
    feat = torch.randn(n, f, generator=g).to(dev)
    w = torch.randn(d, f, generator=g).to(dev)
    b = torch.randn(d, generator=g).to(dev)
    dx0 = torch.randn(n, d, generator=g).to(dev)
    stream = _lib.current_stream(dev)
    x0 = torch.empty(n, d, device=dev)
    _lib.check(lib.rwk_project_forward(feat.data_ptr(), w.data_ptr(), b.data_ptr(), n, f, d, x0.data_ptr(), stream),
               "project_forward")
    fd, wd, bd = feat.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    ref = torch.sigmoid(fd @ wd.t() + bd)
    assert rel_err(x0.cpu().numpy(), ref.detach().cpu().numpy()) < 1e-6
    ref.backward(dx0.double())
    blocks = lib.rwk_project_backward_blocks(n, d)
    partials = torch.empty(blocks, d * f + d, device=dev)
    d_feat = torch.empty(n, f, device=dev)
    _lib.check(lib.rwk_project_backward(dx0.data_ptr(), x0.data_ptr(), feat.data_ptr(), w.data_ptr(), n, f, d,
                                        partials.data_ptr(), d_feat.data_ptr(), stream), "project_backward")
    tot = partials.double().sum(0).cpu().numpy()
    assert rel_err(tot[:d * f].reshape(d, f), wd.grad.cpu().numpy()) < 1e-5
    assert rel_err(tot[d * f:], bd.grad.cpu().numpy()) < 1e-5
    assert rel_err(d_feat.cpu().numpy(), fd.grad.cpu().numpy()) < 1e-5


@pytest.mark.parametrize("gather", [False, True])
def test_flat_adam_matches_torch_adam(gather, dev):
    """gather=True: the gradients are read from the parameters' own .grad tensors inside the Adam launch."""
    from gnn_amd.dp import FlatGradBucket
    from gnn_amd.optim import FlatAdam
    torch.manual_seed(0)
    net_a = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3)).to(dev)
    net_b = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3)).to(dev)
    net_b.load_state_dict(net_a.state_dict())
    opt_a = torch.optim.Adam(net_a.parameters(), lr=1e-2, weight_decay=1e-3)
    bucket = FlatGradBucket(net_b)
    opt_b = FlatAdam(net_b, bucket, lr=1e-2, weight_decay=1e-3)
    x = torch.randn(32, 7, device=dev)
    y = torch.randn(32, 3, device=dev)
    for it in range(25):
        opt_a.zero_grad()
        F.mse_loss(net_a(x), y).backward()
        opt_a.step()
        if gather:
            bucket.release()
            F.mse_loss(net_b(x), y).backward()
            grads = [p.grad.reshape(-1) for p in net_b.parameters()]
            opt_b.step(gather=True)
            assert torch.equal(bucket.flat, torch.cat(grads))
        else:
            opt_b.zero_grad()
            F.mse_loss(net_b(x), y).backward()
            opt_b.step()
    for pa, pb in zip(net_a.parameters(), net_b.parameters()):
        assert rel_err(pb.detach().cpu().numpy(), pa.detach().cpu().numpy()) < 1e-5
    assert int(opt_b.step_counters[0].item()) == 25


def _trainer(c, hp, dev, seed=99, world=1, **kw):
    import bench
    from gnn_amd import RW_NN
    torch.manual_seed(1)
    m = RW_NN(hp["input_dim"], hp["max_step"], hp["hidden_graphs"], hp["size_hidden_graphs"], hp["hidden_dim"],
              hp["penultimate_dim"], bool(hp["normalize"]), hp["n_classes"], hp["dropout"], dev).to(dev)
    m.load_state_dict(c.torch_state())
    m.train()
    m.seed_dropout(seed)
    return bench.Trainer(m, c.torch_inputs(dev), world, **kw)


@pytest.mark.parametrize("world", [1, 2])
def test_graph_replay_trains_like_eager(world, dev):
    """Without dropout the captured step and the eager step follow the same trajectory.  world = 2 takes the
    data-parallel shape of the step (gradient gather, a slot for the all-reduce - a no-op without a process
    group - and the optimiser in a second graph) on one device."""
    c = GoldenCase("mutag_b0")
    hp = dict(c.hp, dropout=0.0)
    eager, graph = _trainer(c, hp, dev, world=world), _trainer(c, hp, dev, world=world)
    assert (graph.g_opt is None)
    graph.capture()                      # runs 3 eager warm-up steps, then captures
    assert graph.g_opt is None and graph.opt_in_graph is False     # the flat Adam launch needs no graph of its own
    for _ in range(3):
        eager.step_eager()
    for _ in range(5):
        eager.step_eager()
        graph.step_graph()
        assert abs(float(eager.loss) - float(graph.loss)) < 1e-4 * abs(float(eager.loss))
    for pe, pg in zip(eager.model.parameters(), graph.model.parameters()):
        assert rel_err(pg.detach().cpu().numpy(), pe.detach().cpu().numpy()) < 1e-3


@pytest.mark.parametrize("drop", [0.0, 0.2])
def test_multi_step_graph_equals_single_step_replays(dev, drop):
    """A graph that holds three consecutive steps (Trainer.capture(steps_per_graph=3), run_steps) against the same
    number of single-step replays: parameters, moments, step counters, running statistics, generator state and the
    losses bit for bit - with dropout too, its generator state advancing on the device inside the graph."""
    c = GoldenCase("mutag_b0")
    hp = dict(c.hp, dropout=drop)
    multi, single = _trainer(c, hp, dev), _trainer(c, hp, dev)
    multi.capture(steps_per_graph=3)
    single.capture()
    assert multi.multi_k == 3 and single.multi_k == 1
    multi.run_steps(8)                    # two replays of three steps, two single replays
    singles = []
    for _ in range(8):
        single.step_graph()
        singles.append(float(single.loss))
    torch.cuda.synchronize()
    assert float(multi.loss) == singles[-1]
    multi.run_steps(3)                    # ends on a multi-step replay
    for _ in range(3):
        single.step_graph()
        singles.append(float(single.loss))
    torch.cuda.synchronize()
    assert [float(v) for v in multi.losses] == singles[-3:] and float(multi.loss) == singles[-1]
    for a, b in ((multi.opt.flat_param, single.opt.flat_param), (multi.opt.exp_avg, single.opt.exp_avg),
                 (multi.opt.exp_avg_sq, single.opt.exp_avg_sq), (multi.opt.step_counters, single.opt.step_counters),
                 (multi.model._rng_state, single.model._rng_state),
                 (multi.model.bn.running_mean, single.model.bn.running_mean),
                 (multi.model.bn.running_var, single.model.bn.running_var),
                 (multi.model.bn.num_batches_tracked, single.model.bn.num_batches_tracked)):
        assert torch.equal(a, b)


@pytest.mark.parametrize("drop", [0.0, 0.2])
def test_head_finish_folded_into_the_layer_backward(dev, drop, monkeypatch):
    """Single-process step without the head's finishing launch (function.FOLD_HEAD_FINISH): the batch-norm backward
    inside the fused backward kernel, the head's parameter-gradient sums and the mean loss inside the step's last
    launch - against the same step with the head's own launch: parameters, moments, flat gradient, the .grad
    tensors, loss and step counters bit for bit, eagerly, replayed from a graph, and when the two shapes alternate."""
    import gnn_amd.function as fn
    c = GoldenCase("mutag_b0")
    hp = dict(c.hp, dropout=drop)
    folded, plain = _trainer(c, hp, dev), _trainer(c, hp, dev)
    assert folded.tail is not None and plain.tail is not None

    def step(tr, fold, graph=False):
        monkeypatch.setattr(fn, "FOLD_HEAD_FINISH", fold)
        tr.step_graph() if graph else tr.step_eager()

    def same():
        torch.cuda.synchronize()
        for a, b in ((folded.opt.flat_param, plain.opt.flat_param), (folded.opt.exp_avg, plain.opt.exp_avg),
                     (folded.opt.exp_avg_sq, plain.opt.exp_avg_sq), (folded.bucket.flat, plain.bucket.flat),
                     (folded.model._rng_state, plain.model._rng_state)):
            assert torch.equal(a, b)
        for pf, pp in zip(folded.model.parameters(), plain.model.parameters()):
            assert torch.equal(pf.grad, pp.grad)
        assert float(folded.loss) == float(plain.loss)

    for _ in range(3):
        step(folded, True)
        step(plain, False)
        same()
    assert folded.tail.head_folds == 3 and plain.tail.head_folds == 0
    t = int(folded.opt.step_counters[0])
    assert int(folded.opt.step_counters.min()) == t == 3
    step(folded, False)                   # the shapes alternate: the counters follow
    step(plain, False)
    step(folded, True)
    step(plain, False)
    same()
    monkeypatch.setattr(fn, "FOLD_HEAD_FINISH", True)
    folded.capture()                      # 3 eager steps, then the captured one
    monkeypatch.setattr(fn, "FOLD_HEAD_FINISH", False)
    plain.capture()
    n = folded.tail.head_folds
    for _ in range(3):
        step(folded, True, graph=True)
        step(plain, False, graph=True)
    same()
    assert folded.tail.head_folds == n    # replays launch nothing from the host
    assert int(folded.opt.step_counters.min()) == int(folded.opt.step_counters.max()) == int(plain.opt.step_counters[0])


@pytest.mark.parametrize("drop", [0.0, 0.2])
def test_data_parallel_step_shape_without_gather_and_head_finish_launches(dev, drop):
    """The step in the shape of a 2-rank run (no process group: the all-reduce slot is a no-op): the backward pass's
    last launch completes the flat gradient bucket - no gather launch, no finishing launch of the head, no update -
    against the same shape with those launches: bucket, parameters, moments and loss bit for bit, eager and replayed."""
    c = GoldenCase("mutag_b0")
    hp = dict(c.hp, dropout=drop)
    merged, plain = _trainer(c, hp, dev, world=2), _trainer(c, hp, dev, world=2, merge_tail=False)
    assert merged.tail is not None and not merged.tail.update and plain.tail is None

    def same():
        torch.cuda.synchronize()
        for a, b in ((merged.opt.flat_param, plain.opt.flat_param), (merged.opt.exp_avg, plain.opt.exp_avg),
                     (merged.opt.exp_avg_sq, plain.opt.exp_avg_sq), (merged.bucket.flat, plain.bucket.flat)):
            assert torch.equal(a, b)
        assert float(merged.loss) == float(plain.loss)

    for _ in range(3):
        merged.step_eager()
        plain.step_eager()
        same()
    assert merged.tail.head_folds == 3
    merged.capture()
    plain.capture()
    assert merged.g_opt is None and not merged.collective_captured
    for _ in range(3):
        merged.step_graph()
        plain.step_graph()
    same()


@pytest.mark.parametrize("drop", [0.0, 0.2])
def test_optimiser_update_inside_the_last_backward_launch(dev, drop):
    """Single-process step: the Adam update applied by the backward pass's last launch (optim.StepTail) against the
    same step with the stand-alone Adam launch - parameters, both moment buffers, the flat gradient and the step
    counters bit for bit after several steps, eagerly and replayed from a graph; the hand-over is armed only inside
    the trainer's own step."""
    c = GoldenCase("mutag_b0")
    hp = dict(c.hp, dropout=drop)
    merged, plain = _trainer(c, hp, dev), _trainer(c, hp, dev, merge_tail=False)
    assert merged.tail is not None and plain.tail is None
    for _ in range(4):
        merged.step_eager()
        plain.step_eager()
        assert not merged.tail.done and not merged.tail.armed
    assert int(merged.opt.step_counters.min()) == 4 and int(merged.opt.step_counters.max()) == 4
    for a, b in ((merged.opt.flat_param, plain.opt.flat_param), (merged.opt.exp_avg, plain.opt.exp_avg),
                 (merged.opt.exp_avg_sq, plain.opt.exp_avg_sq), (merged.bucket.flat, plain.bucket.flat)):
        assert torch.equal(a, b)
    assert float(merged.loss) == float(plain.loss)
    merged.capture()                      # 3 more eager steps, then the captured one
    for _ in range(3):
        plain.step_eager()
    for _ in range(3):
        merged.step_graph()
        plain.step_eager()
    torch.cuda.synchronize()
    assert int(merged.opt.step_counters.min()) == 10 and int(merged.opt.step_counters.max()) == 10
    assert torch.equal(merged.opt.flat_param, plain.opt.flat_param)
    assert torch.equal(merged.opt.exp_avg_sq, plain.opt.exp_avg_sq)
    # outside the trainer's step the model's backward leaves the parameters alone
    before = merged.opt.flat_param.clone()
    _, loss = merged.model.forward_with_loss(merged.adj, merged.feats, merged.gi, merged.y)
    loss.backward()
    torch.cuda.synchronize()
    assert torch.equal(merged.opt.flat_param, before)


def test_data_parallel_step_with_captured_collective_slot(dev):
    """The data-parallel step with the all-reduce inside the step's graph (one replay per step): on one device
    without a process group the collective is a no-op, which exercises the capture, its self-check and the replay;
    the trajectory must be the one of the two-graph shape.  Also the per-phase timing used by the benchmark."""
    c = GoldenCase("mutag_b0")
    hp = dict(c.hp, dropout=0.0)
    two, one = _trainer(c, hp, dev, world=2), _trainer(c, hp, dev, world=2, capture_collective=True)
    two.capture()
    one.capture()
    assert one.collective_captured and one.g_opt is None and "captured" in one.capture_note
    assert not two.collective_captured and two.g_opt is None and not two.opt_in_graph
    for _ in range(4):
        two.step_graph()
        one.step_graph()
    torch.cuda.synchronize()
    # `one` ran one step more (the replay of its self-check)
    two.step_graph()
    torch.cuda.synchronize()
    for pa, pb in zip(two.model.parameters(), one.model.parameters()):
        assert torch.equal(pa, pb)
    ph = two.phase_times(steps=5)
    assert set(ph) == {"step_device_ms", "step_host_sync_ms", "fwd_bwd_graph_ms", "all_reduce_ms", "optimizer_ms"}
    assert all(v >= 0 for v in ph.values())
    assert set(one.phase_times(steps=3)) == {"step_device_ms", "step_host_sync_ms"}


def test_graph_replay_draws_fresh_dropout_masks(dev):
    """The dropout offset lives on the device: every replay of the captured step advances it."""
    c = GoldenCase("mutag_b0")
    # captured in the mode the data-parallel runs use (the watchdog thread of the collective library is alive then)
    graph = _trainer(c, dict(c.hp, dropout=0.2), dev, capture_error_mode="thread_local")
    graph.capture()
    offsets, losses = [], []
    for _ in range(5):
        graph.step_graph()
        offsets.append(int(graph.model._rng_state[1].item()))
        losses.append(float(graph.loss))
    # two masks per step: the random-walk layer's and the head's, one generator offset each
    assert offsets == list(range(offsets[0], offsets[0] + 10, 2))
    assert all(np.isfinite(losses)) and losses[-1] < losses[0] + 0.2


def test_fused_cross_entropy_matches_torch(dev):
    """gnn_amd.losses.cross_entropy against F.cross_entropy in float64 (value and gradient), on log-probabilities
    as main.py:85 feeds it and on raw scores."""
    from gnn_amd.losses import cross_entropy
    rng = np.random.default_rng(9)
    for g, nc, logp in ((1024, 2, True), (7, 5, False), (5000, 10, True), (1, 3, False)):
        x = rng.normal(size=(g, nc)).astype(np.float32) * 3
        if logp:
            x = (torch.log_softmax(torch.from_numpy(x), dim=1)).numpy()
        y = rng.integers(0, nc, size=g)
        x64 = torch.from_numpy(x).double().requires_grad_(True)
        ref = F.cross_entropy(x64, torch.from_numpy(y))
        (ref * 1.7).backward()
        xd = torch.from_numpy(x).to(dev).requires_grad_(True)
        got = cross_entropy(xd, torch.from_numpy(y).to(dev))
        (got * 1.7).backward()
        assert got.shape == () and abs(float(got) - float(ref)) <= 1e-5 * max(abs(float(ref)), 1e-3)
        assert rel_err(xd.grad.cpu().numpy(), x64.grad.numpy()) < 1e-5


def test_structure_cache_entry_lives_as_long_as_its_tensors(dev):
    """The per-batch structure is cached by tensor identity and released with the batch: a training script that
    builds new batches for every fold does not accumulate the CSR arrays of the finished ones."""
    import gc
    from gnn_amd import get_structure
    from gnn_amd.structure import _CACHE, _key
    c = GoldenCase("mutag_b0")
    adj, feats, gi, _ = c.torch_inputs(dev)
    key = _key(adj, gi)
    st = get_structure(adj, gi)
    assert get_structure(adj, gi) is st and key in _CACHE.entries
    del st, adj
    gc.collect()
    assert key not in _CACHE.entries
    adj2, _, gi2, _ = c.torch_inputs(dev)
    key2 = _key(adj2, gi2)
    st2 = get_structure(adj2, gi2)
    assert key2 in _CACHE.entries
    del gi2
    gc.collect()
    assert key2 not in _CACHE.entries
    assert st2.n == c.n_nodes


def test_fused_loss_is_nan_until_backward_and_labels_are_checked(dev):
    """model.forward_with_loss in training mode: the loss value is finished by the backward launch and reads NaN
    (never stale memory) before it; a label outside [0, n_classes) raises as F.cross_entropy does."""
    c = GoldenCase("mutag_b0")
    tr = _trainer(c, dict(c.hp, dropout=0.2), dev)
    m = tr.model
    adj, feats, gi, y = c.torch_inputs(dev)
    out, loss = m.forward_with_loss(adj, feats, gi, y)
    torch.cuda.synchronize()
    assert bool(torch.isnan(loss.detach()))
    loss.backward()
    torch.cuda.synchronize()
    ref = F.cross_entropy(out.detach(), y)
    assert abs(float(loss) - float(ref)) <= 1e-5 * abs(float(ref))
    bad = y.clone()
    bad[3] = c.hp["n_classes"]
    with pytest.raises(IndexError):
        m.forward_with_loss(adj, feats, gi, bad)
    bad[3] = -100
    with pytest.raises(IndexError):
        m.forward_with_loss(adj, feats, gi, bad)


def test_capture_refuses_a_leaked_autograd_graph(dev):
    """A reference to an earlier step's loss keeps that step's autograd graph, and with it AccumulateGrad nodes bound
    to the stream it ran on, alive; a backward pass captured in that state is forked over two streams (ending such a
    capture has crashed the process).  Trainer.capture reports it as a RuntimeError from its first warm-up step,
    before any capture begins; once the reference is gone the capture goes through."""
    c = GoldenCase("mutag_b0")
    tr = _trainer(c, dict(c.hp, dropout=0.0), dev)
    tr.step_eager()
    leaked = tr.loss
    assert leaked.grad_fn is not None
    with pytest.raises(RuntimeError, match="still alive"):
        tr.capture()
    assert tr.g_fb is None
    del leaked
    tr.capture()
    tr.step_graph()
    torch.cuda.synchronize()
    assert np.isfinite(float(tr.loss))
