"""``gnn_amd.Adam`` - the one-launch, torch.optim-compatible optimiser for the reference's loop
(rwgnn2/main.py:79-88,108,127-135,143-146) - against ``torch.optim.Adam`` on the same model and batches: trajectories,
a StepLR schedule, and a checkpoint round trip in torch's own state_dict layout."""
import copy

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from _golden import rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _setup(dev, opt_cls, **kw):
    from gnn_amd import RW_NN
    torch.manual_seed(0)
    m = RW_NN(1, 2, 16, 5, 4, 32, False, 2, 0.0, dev).to(dev).train()      # dropout 0: identical gradients
    return m, opt_cls(m.parameters(), lr=1e-2, **kw)


def _train(m, opt, batch, sched=None):
    adj, feats, gi, y = batch
    opt.zero_grad()
    loss = F.cross_entropy(m(adj, feats, gi), y)
    loss.backward()
    opt.step()
    if sched is not None:
        sched.step()
    return float(loss)


@pytest.mark.parametrize("weight_decay", [0.0, 1e-2])
def test_one_launch_adam_follows_torch_adam(dev, weight_decay):
    import gnn_amd
    from gnn_amd.synth import random_batch
    batch = random_batch(40, 5, 40, 3.0, seed=1).to_torch(dev)
    m_ref, o_ref = _setup(dev, torch.optim.Adam, weight_decay=weight_decay)
    m_new, o_new = _setup(dev, gnn_amd.Adam, weight_decay=weight_decay)
    s_ref = torch.optim.lr_scheduler.StepLR(o_ref, step_size=4, gamma=0.5)          # main.py:80
    s_new = torch.optim.lr_scheduler.StepLR(o_new, step_size=4, gamma=0.5)
    for step in range(10):
        a, b = _train(m_ref, o_ref, batch, s_ref), _train(m_new, o_new, batch, s_new)
        assert abs(a - b) <= 2e-5 * max(1.0, abs(a)), step
    assert o_new.param_groups[0]["lr"] == o_ref.param_groups[0]["lr"] == 1e-2 * 0.25
    for (n1, p1), (_, p2) in zip(m_ref.named_parameters(), m_new.named_parameters()):
        assert rel_err(p2.detach().cpu().numpy(), p1.detach().cpu().numpy()) < 2e-4, n1


def test_checkpoint_round_trip_in_torch_layout(dev):
    """main.py:127-135,143-146: torch.save({'state_dict': ..., 'optimizer': optimizer.state_dict()}) and back, also
    across the two optimiser classes (the layout is torch.optim.Adam's)."""
    import gnn_amd
    from gnn_amd.synth import random_batch
    batch = random_batch(30, 5, 30, 3.0, seed=2).to_torch(dev)
    m, o = _setup(dev, gnn_amd.Adam)
    for _ in range(3):
        _train(m, o, batch)
    sd_model, sd_opt = copy.deepcopy(m.state_dict()), copy.deepcopy(o.state_dict())
    assert sorted(sd_opt["state"][0]) == ["exp_avg", "exp_avg_sq", "step"] and float(sd_opt["state"][0]["step"]) == 3
    cont = [_train(m, o, batch) for _ in range(3)]
    # a fresh model + gnn_amd.Adam from the checkpoint continues identically
    m2, o2 = _setup(dev, gnn_amd.Adam)
    m2.load_state_dict(sd_model)
    o2.load_state_dict(sd_opt)
    assert [_train(m2, o2, batch) for _ in range(3)] == cont
    # torch.optim.Adam accepts the same checkpoint and follows to rounding
    m3, o3 = _setup(dev, torch.optim.Adam)
    m3.load_state_dict(sd_model)
    o3.load_state_dict(sd_opt)
    got = [_train(m3, o3, batch) for _ in range(3)]
    assert np.allclose(got, cont, rtol=2e-5, atol=1e-6)
