"""Two-hop neighbourhood encoding (rwgnn2/rwgnn_dis_bias6.py:531-561) on the SpMM kernels against a float64
dense-matrix evaluation of the reference's formula (A of ones over the valid edges, duplicates added).  Parity is
unpinned: the module that holds the reference function imports torch_geometric at its top and cannot be loaded
here, and the reference holds no fixtures for it."""
import numpy as np
import pytest
import torch

from _golden import rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


@pytest.mark.parametrize("n,e,c,max_step", [(60, 200, 8, 2), (500, 1500, 16, 1), (2000, 9000, 64, 3), (7, 0, 4, 2)])
def test_neighborhood_encoding(n, e, c, max_step, dev):
    from gnn_amd.neighborhood import neighborhood_encoding
    rng = np.random.default_rng(n)
    ei = rng.integers(0, n, size=(2, e))
    if e:
        ei = np.concatenate([ei, ei[:, :5], np.array([[1, n + 3], [n, 2]])], axis=1)   # duplicates, invalid end points
    x = rng.normal(size=(n, c)).astype(np.float32)
    d_out = rng.normal(size=(n, c)).astype(np.float32)
    xt = torch.from_numpy(x).to(dev).requires_grad_(True)
    out = neighborhood_encoding(xt, torch.from_numpy(ei.astype(np.int64)).to(dev), max_step)
    out.backward(torch.from_numpy(d_out).to(dev))
    a = np.zeros((n, n))
    for r, cc in ei.T:
        if r < n and cc < n:
            a[r, cc] += 1.0
    x64 = x.astype(np.float64)
    if e == 0:
        want, d_want = x64, d_out.astype(np.float64)
    elif max_step > 1:
        m = (np.eye(n) + a + 0.5 * a @ a) / 2.5
        want, d_want = m @ x64, m.T @ d_out
    else:
        m = (np.eye(n) + a) / 2.0
        want, d_want = m @ x64, m.T @ d_out
    assert rel_err(out.detach().cpu().numpy(), want) < 1e-5
    assert rel_err(xt.grad.cpu().numpy(), d_want) < 1e-5
