"""RWConv (row-normalised propagation + Linear, rwgnn/rwgnn_pyg.py:19-46) on the SpMM engine against the float64
restatement in oracle/rwconv_oracle.py.  Parity is unpinned for this layer: torch_geometric, on which the
reference builds it, is not available, and the reference holds no fixtures for it; the oracle restates the
library's published semantics."""
import numpy as np
import pytest
import torch

from _golden import rel_err
from oracle import rwconv_oracle as R

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _edges(rng, n, e, directed):
    src = rng.integers(0, n, size=e)
    dst = rng.integers(0, n, size=e)
    if not directed:
        src, dst = np.concatenate([src, dst]), np.concatenate([dst, src])
    # duplicates, explicit self-loops and a node without any edge (n - 1) on purpose
    src = np.concatenate([src, src[:7], [3, 3, 5]])
    dst = np.concatenate([dst, dst[:7], [3, 3, 5]])
    keep = (src != n - 1) & (dst != n - 1)
    return np.vstack([src[keep], dst[keep]]).astype(np.int64)


@pytest.mark.parametrize("n,e,c_in,c_out,directed", [(50, 120, 4, 4, False), (1000, 4000, 4, 4, True),
                                                      (333, 900, 16, 7, True), (4000, 9000, 64, 32, False),
                                                      (10, 0, 3, 5, True)])
def test_rwconv_forward_backward(n, e, c_in, c_out, directed, dev):
    from gnn_amd.rwconv import RWConv
    rng = np.random.default_rng(n + e)
    ei = _edges(rng, n, e, directed)
    x = rng.normal(size=(n, c_in)).astype(np.float32)
    d_out = rng.normal(size=(n, c_out)).astype(np.float32)
    torch.manual_seed(n)
    layer = RWConv(c_in, c_out).to(dev)
    w = layer.lin.weight.detach().cpu().numpy()
    b = layer.lin.bias.detach().cpu().numpy()
    xt = torch.from_numpy(x).to(dev).requires_grad_(True)
    eit = torch.from_numpy(ei).to(dev)
    out = layer(xt, eit)
    out.backward(torch.from_numpy(d_out).to(dev))
    want, _ = R.rwconv_forward(x, ei, w, b)
    d_x, d_w, d_b = R.rwconv_backward(x, ei, w, d_out)
    assert rel_err(out.detach().cpu().numpy(), want) < TOL
    assert rel_err(xt.grad.cpu().numpy(), d_x) < TOL
    assert rel_err(layer.lin.weight.grad.cpu().numpy(), d_w) < TOL
    assert rel_err(layer.lin.bias.grad.cpu().numpy(), d_b) < TOL
    # second call: the structure comes from the cache, the result is the same bits
    out2 = layer(xt.detach(), eit)
    assert torch.equal(out2, out.detach())


def test_rwconv_rejects_host_tensors():
    from gnn_amd.rwconv import RWConv
    layer = RWConv(4, 4)
    with pytest.raises(RuntimeError):
        layer(torch.zeros(5, 4), torch.zeros((2, 3), dtype=torch.int64))
