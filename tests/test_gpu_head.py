"""Fused classifier head (rwgnn2/model.py:68-72) against the torch modules evaluated in float64 on the CPU.

Tolerance: 2e-5 of the largest reference magnitude of each tensor (fp32 arithmetic against an fp64 reference;
batch-norm divides by a standard deviation, which amplifies rounding of the inputs a little beyond 1e-5).
"""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = 2e-5


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def rel(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-30))


class Head(nn.Module):
    def __init__(self, c, p, nc):
        super().__init__()
        self.bn = nn.BatchNorm1d(c)
        self.fc1 = nn.Linear(c, p)
        self.fc2 = nn.Linear(p, nc)


def make(c, p, nc, seed):
    torch.manual_seed(seed)
    m = Head(c, p, nc)
    with torch.no_grad():
        m.bn.weight.uniform_(0.5, 1.5)
        m.bn.bias.uniform_(-0.5, 0.5)
        m.bn.running_mean.uniform_(-1, 1)
        m.bn.running_var.uniform_(0.5, 2.0)
    return m


def make_case(g, c, p, nc, training):
    """Module pair (fp32, fp64), inputs and labels.  ReLU is discontinuous in its gradient: a pre-activation
    within fp32 rounding of zero can fall on either side in fp32 and fp64, which changes gradients by a whole
    term, so cases with such a pre-activation are re-drawn."""
    for attempt in range(50):
        seed = 1000 * attempt + g + c
        m = make(c, p, nc, seed=seed)
        m64 = make(c, p, nc, seed=seed).double()
        rng = np.random.default_rng(seed)
        # random-walk-kernel features are positive with a large common offset: a hard case for the variance
        k = torch.from_numpy((rng.uniform(0, 1, size=(g, c)) * 3 + 50).astype(np.float32))
        y = torch.from_numpy(rng.integers(0, nc, size=g))
        with torch.no_grad():
            m64.train(training)
            rm, rv = m64.bn.running_mean.clone(), m64.bn.running_var.clone()
            pre = m64.fc1(m64.bn(k.double()))
            m64.bn.running_mean.copy_(rm)
            m64.bn.running_var.copy_(rv)
            m64.bn.num_batches_tracked.zero_()
        if float(pre.abs().min()) > 4e-6:
            return m, m64, k, y
    raise AssertionError("no well-separated case found")


def reference(m64, k64, y, training, mask=None, scale=1.0):
    m64.train(training)
    h = F.relu(m64.fc1(m64.bn(k64)))
    if mask is not None:
        h = h * mask * scale
    out = F.log_softmax(m64.fc2(h), dim=1)
    loss = F.cross_entropy(out, y)
    return out, loss


@pytest.mark.parametrize("g,c,p,nc", [(64, 32, 32, 2), (33, 32, 32, 2), (1000, 128, 32, 7), (9000, 48, 20, 3),
                                      (2, 16, 8, 2), (257, 256, 16, 5), (700, 160, 32, 2), (300, 256, 32, 3)])
@pytest.mark.parametrize("training", [True, False])
def test_head_matches_torch_fp64(dev, g, c, p, nc, training):
    from gnn_amd.head import fused_head, head_supported
    assert head_supported(g, c, p, nc)
    m, m64, k, y = make_case(g, c, p, nc, training)
    k64 = k.double().requires_grad_(True)
    out_ref, loss_ref = reference(m64, k64, y, training)
    loss_ref.backward()

    md = m.to(dev)
    md.train(training)
    kd = k.to(dev).requires_grad_(True)
    out = fused_head(kd, md.bn, md.fc1, md.fc2, 0.0, training)
    loss = F.cross_entropy(out, y.to(dev))
    loss.backward()
    assert rel(out, out_ref) < TOL
    assert abs(float(loss) - float(loss_ref)) < TOL * abs(float(loss_ref))
    # sums over up to 9000 rows of terms of either sign: torch's own fp32 kernels reach 1e-5 .. 5e-5 here
    assert rel(kd.grad, k64.grad) < 5e-5
    for (name, prm), (_, ref) in zip(md.named_parameters(), m64.named_parameters()):
        assert rel(prm.grad, ref.grad) < 1e-4, name
    assert rel(md.bn.running_mean, m64.bn.running_mean) < TOL
    assert rel(md.bn.running_var, m64.bn.running_var) < 1e-4
    assert int(md.bn.num_batches_tracked) == int(m64.bn.num_batches_tracked)


def test_head_gradients_well_conditioned(dev):
    """Centred inputs: every gradient, batch-norm's included, to 2e-5."""
    from gnn_amd.head import fused_head
    g, c, p, nc = 512, 32, 32, 2
    m, m64 = make(c, p, nc, 3), make(c, p, nc, 3).double()
    rng = np.random.default_rng(11)
    k = torch.from_numpy(rng.normal(size=(g, c)).astype(np.float32))
    y = torch.from_numpy(rng.integers(0, nc, size=g))
    k64 = k.double().requires_grad_(True)
    out_ref, loss_ref = reference(m64, k64, y, True)
    loss_ref.backward()
    md = m.to(dev).train()
    kd = k.to(dev).requires_grad_(True)
    out = fused_head(kd, md.bn, md.fc1, md.fc2, 0.0, True)
    F.cross_entropy(out, y.to(dev)).backward()
    assert rel(out, out_ref) < TOL
    assert rel(kd.grad, k64.grad) < TOL
    for (name, prm), (_, ref) in zip(md.named_parameters(), m64.named_parameters()):
        assert rel(prm.grad, ref.grad) < TOL, name


def test_head_dropout(dev):
    """Keep rate, exact 1/(1-p) scaling, gradients consistent with the drawn mask, fresh mask per call,
    bitwise reproducibility for a fixed generator state."""
    from gnn_amd.head import fused_head
    g, c, p, nc, drop = 4096, 32, 32, 2, 0.3
    m, m64 = make(c, p, nc, 5), make(c, p, nc, 5).double()
    rng = np.random.default_rng(2)
    k = torch.from_numpy(rng.normal(size=(g, c)).astype(np.float32))
    y = torch.from_numpy(rng.integers(0, nc, size=g))
    md = m.to(dev).train()
    state = torch.tensor([777, 5], dtype=torch.int64, device=dev)
    kd = k.to(dev).requires_grad_(True)
    out = fused_head(kd, md.bn, md.fc1, md.fc2, drop, True, state)
    assert state.tolist() == [777, 6]
    h1d = out.grad_fn.saved_tensors[7]
    # reference hidden activations without dropout
    k64 = k.double().requires_grad_(True)
    m64.train()
    h_ref = F.relu(m64.fc1(m64.bn(k64)))
    active = h_ref.detach() > 1e-6
    kept = (h1d.cpu() != 0) & active
    rate = float(kept.sum()) / float(active.sum())
    assert abs(rate - (1 - drop)) < 4 * np.sqrt(drop * (1 - drop) / float(active.sum()))
    mask = torch.where(active, kept, torch.ones_like(kept)).double()
    m64b = make(c, p, nc, 5).double()
    k64b = k.double().requires_grad_(True)
    out_ref, loss_ref = reference(m64b, k64b, y, True, mask=mask, scale=1.0 / (1.0 - drop))
    loss_ref.backward()
    F.cross_entropy(out, y.to(dev)).backward()
    assert rel(out, out_ref) < TOL
    assert rel(kd.grad, k64b.grad) < TOL
    for (name, prm), (_, ref) in zip(md.named_parameters(), m64b.named_parameters()):
        assert rel(prm.grad, ref.grad) < TOL, name
    # same state -> same bits; advanced state -> another mask
    state2 = torch.tensor([777, 5], dtype=torch.int64, device=dev)
    out2 = fused_head(k.to(dev), md.bn, md.fc1, md.fc2, drop, True, state2)
    assert torch.equal(out2, out.detach())
    out3 = fused_head(k.to(dev), md.bn, md.fc1, md.fc2, drop, True, state2)
    assert not torch.equal(out3, out2)


def test_head_limits_and_errors(dev):
    from gnn_amd.head import fused_head, head_supported
    assert not head_supported(10, 300, 8, 2)
    assert not head_supported(10, 256, 33, 2)          # penultimate_dim * features > 8192
    assert not head_supported(10, 32, 32, 40)
    m = make(8, 4, 2, 0).to(dev).train()
    with pytest.raises(ValueError):
        fused_head(torch.zeros(1, 8, device=dev), m.bn, m.fc1, m.fc2, 0.0, True)


def test_model_head_switch(dev):
    """The model gives the same result through the fused head and through the torch modules (no dropout)."""
    from gnn_amd import RW_NN
    from gnn_amd.synth import random_batch
    b = random_batch(48, 10, 30, 2.5, seed=3)
    adj, feats, gi, y = b.to_torch(dev)
    torch.manual_seed(1)
    model = RW_NN(1, 2, 16, 5, 4, 32, False, 2, 0.0, dev).to(dev).train()
    res = {}
    for fused in (True, False):
        model.fused_head = fused
        model.zero_grad()
        state = {k_: v.clone() for k_, v in model.state_dict().items()}
        out = model(adj, feats, gi)
        F.cross_entropy(out, y).backward()
        res[fused] = (out.detach().clone(), {n: p_.grad.clone() for n, p_ in model.named_parameters()},
                      model.bn.running_var.clone())
        model.load_state_dict(state)
    assert rel(res[True][0], res[False][0]) < 1e-4
    for n in res[True][1]:
        assert rel(res[True][1][n], res[False][1][n]) < 5e-3, n
    assert rel(res[True][2], res[False][2]) < 1e-4


@pytest.mark.parametrize("g,c,p,nc", [(64, 32, 32, 2), (33, 32, 32, 2), (1000, 128, 32, 7), (9000, 48, 20, 3),
                                      (2, 16, 8, 2), (1024, 32, 32, 2), (4096, 128, 32, 2)])
def test_head_with_loss_equals_head_then_cross_entropy(dev, g, c, p, nc):
    """fused_head_loss (the cross-entropy inside the head's launches) against fused_head followed by
    gnn_amd.losses.cross_entropy: same log-probabilities bit for bit, the same gradients (the loss gradient is formed
    by the same expressions, in place: equal bits expected, 1e-6 allowed), the loss value to 1e-6 (its rows are added
    in another order), and against the float64 torch reference at the tolerances of the separate path.  At 32 x 32
    with at most four classes the training step runs the one-wavefront-per-tile kernel, whose inner products are
    added in another order: log-probabilities to 2e-6 and gradients to 2e-5 of the separate path there."""
    from gnn_amd.head import fused_head, fused_head_loss
    from gnn_amd.losses import cross_entropy
    m, m64, k, y = make_case(g, c, p, nc, True)
    k64 = k.double().requires_grad_(True)
    out_ref, loss_ref = reference(m64, k64, y, True)
    loss_ref.backward()

    import copy
    ma = m.to(dev).train()
    mb = copy.deepcopy(ma)
    yd = y.to(dev)
    ka = k.to(dev).requires_grad_(True)
    out_a = fused_head(ka, ma.bn, ma.fc1, ma.fc2, 0.0, True)
    loss_a = cross_entropy(out_a, yd)
    loss_a.backward()
    kb = k.to(dev).requires_grad_(True)
    out_b, loss_b = fused_head_loss(kb, yd, mb.bn, mb.fc1, mb.fc2, 0.0, True)
    assert not out_b.requires_grad
    loss_b.backward()
    torch.cuda.synchronize()
    one_wave = c == 32 and p == 32 and nc <= 4
    if one_wave:
        assert rel(out_b, out_a.detach()) < 2e-6
    else:
        assert torch.equal(out_a.detach(), out_b)
    same = 2e-5 if one_wave else 1e-6
    assert abs(float(loss_a) - float(loss_b)) < 1e-6 * abs(float(loss_a))
    assert abs(float(loss_b) - float(loss_ref)) < TOL * abs(float(loss_ref))
    assert rel(kb.grad, ka.grad) < same
    assert rel(kb.grad, k64.grad) < 5e-5
    for (name, pa), (_, pb), (_, ref) in zip(ma.named_parameters(), mb.named_parameters(), m64.named_parameters()):
        assert rel(pb.grad, pa.grad) < same, name
        assert rel(pb.grad, ref.grad) < 1e-4, name
    assert torch.equal(ma.bn.running_mean, mb.bn.running_mean)
    assert torch.equal(ma.bn.running_var, mb.bn.running_var)


def test_head_with_loss_scaled_gradient_and_no_grad(dev):
    """d_loss other than one scales every gradient; without gradients the loss is complete at once."""
    from gnn_amd.head import fused_head_loss
    g, c, p, nc = 300, 32, 32, 3
    m, m64, k, y = make_case(g, c, p, nc, True)
    md = m.to(dev).train()
    yd = y.to(dev)
    k1 = k.to(dev).requires_grad_(True)
    _, l1 = fused_head_loss(k1, yd, md.bn, md.fc1, md.fc2, 0.0, True)
    l1.backward()
    g1 = k1.grad.clone()
    k2 = k.to(dev).requires_grad_(True)
    _, l2 = fused_head_loss(k2, yd, md.bn, md.fc1, md.fc2, 0.0, True)
    (l2 * 3.0).backward()
    assert rel(k2.grad, 3.0 * g1) < 1e-6
    md.eval()
    with torch.no_grad():
        out, loss = fused_head_loss(k.to(dev), yd, md.bn, md.fc1, md.fc2, 0.0, False)
    want = F.cross_entropy(out.double(), yd)
    assert abs(float(loss) - float(want)) < 1e-6 * abs(float(want))


def test_model_forward_with_loss(dev):
    """RW_NN.forward_with_loss against forward + F.cross_entropy on the same dropout draws."""
    from gnn_amd import RW_NN
    from gnn_amd.synth import random_batch
    b = random_batch(48, 10, 30, 2.5, seed=5)
    adj, feats, gi, y = b.to_torch(dev)
    torch.manual_seed(0)
    ma = RW_NN(1, 2, 16, 5, 4, 32, False, 2, 0.2, dev).to(dev).train()
    import copy
    mb = copy.deepcopy(ma)
    ma.seed_dropout(77)
    mb.seed_dropout(77)
    out_a = ma(adj, feats, gi)
    loss_a = F.cross_entropy(out_a, y)
    loss_a.backward()
    out_b, loss_b = mb.forward_with_loss(adj, feats, gi, y)
    loss_b.backward()
    torch.cuda.synchronize()
    # 32 x 32 head: the training step's one-wavefront kernel adds the inner products in another order
    assert rel(out_b, out_a.detach()) < 2e-6
    assert abs(float(loss_a) - float(loss_b)) < 2e-6 * abs(float(loss_a))
    for (name, pa), (_, pb) in zip(ma.named_parameters(), mb.named_parameters()):
        assert rel(pb.grad, pa.grad) < 2e-5, name


@pytest.mark.parametrize("drop", [0.0, 0.2])
@pytest.mark.parametrize("with_loss", [False, True])
def test_statistics_from_k_reduction_equal_head_statistics(dev, drop, with_loss):
    """Training mode: the random-walk layer's K reduction hands the head its batch-statistics partials and draws the
    head's dropout offset (one launch less).  Against the same model with that handoff switched off: the same
    generator offsets, hence the same keep masks; K is the same bits, the statistics are merged from other row blocks,
    so outputs, running statistics and gradients agree to float32 rounding of the batch-norm (1e-5 / 2e-4 relative)."""
    import copy
    from gnn_amd import RW_NN
    from gnn_amd.synth import random_batch
    b = random_batch(200, 10, 40, 2.5, seed=11)
    adj, feats, gi, y = b.to_torch(dev)
    torch.manual_seed(3)
    ma = RW_NN(1, 3, 16, 5, 4, 32, False, 2, drop, dev).to(dev).train()
    mb = copy.deepcopy(ma)
    mb._head_handoff = lambda structure, drop_: None
    res = []
    for m in (ma, mb):
        m.seed_dropout(123)
        if with_loss:
            out, loss = m.forward_with_loss(adj, feats, gi, y)
        else:
            out = m(adj, feats, gi)
            loss = F.cross_entropy(out, y)
        loss.backward()
        torch.cuda.synchronize()
        res.append((out.detach(), float(loss), m._rng_state.clone() if drop else None))
    assert ma._head_handoff(type("S", (), {"n_graphs": 200})(), drop) is not None
    assert rel(res[0][0], res[1][0]) < 1e-5
    assert abs(res[0][1] - res[1][1]) < 1e-5 * abs(res[1][1])
    if drop:
        assert torch.equal(res[0][2], res[1][2])          # both paths consumed the same number of draws
    for (name, pa), (_, pb) in zip(ma.named_parameters(), mb.named_parameters()):
        assert rel(pa.grad, pb.grad) < 2e-4, name
    assert rel(ma.bn.running_mean, mb.bn.running_mean) < 1e-6
    assert rel(ma.bn.running_var, mb.bn.running_var) < 1e-5
    assert int(ma.bn.num_batches_tracked) == int(mb.bn.num_batches_tracked) == 1
