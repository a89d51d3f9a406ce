"""Distance-biased transition operator (gnn_amd.distance_bias) against the dense CPU restatement
(oracle/distance_bias_oracle.py) of rwgnn2/rwgnn_dis_bias6.py:109-128, :222-250.  PARITY UNPINNED: the reference
module cannot be imported (torch_geometric) and holds no fixture for these matrices."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from oracle import distance_bias_oracle as oracle  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    return torch.device("cuda:0")


def dense_adj(b):
    a = np.zeros((b.n_nodes, b.n_nodes), dtype=np.float32)
    a[b.indices[0], b.indices[1]] = b.values
    return a


def batches():
    from gnn_amd.synth import powerlaw_batch, random_batch
    return {
        "sparse": random_batch(12, 5, 40, 2.2, seed=1),          # several components per graph: many far pairs
        "denser": random_batch(6, 20, 70, 5.0, seed=2),
        "hubs": powerlaw_batch(5, 30, 90, 2.1, seed=3),
        "one_graph": random_batch(1, 300, 300, 3.0, seed=4),      # bitmap of ten words, rows of hundreds of pairs
        "tiny": random_batch(7, 1, 3, 1.0, seed=5),               # graphs of one to three nodes
    }


@pytest.mark.parametrize("name", ["sparse", "denser", "hubs", "one_graph", "tiny"])
@pytest.mark.parametrize("max_distance", [1, 3, 4])
def test_hop_structure_is_exact(dev, name, max_distance):
    """Integer work, bit-exact: stored pairs = exactly the pairs at 1..max_distance hops with their distance, rows
    ordered by (distance, node), per-level counts."""
    from gnn_amd.distance_bias import distance_bias_structure
    b = batches()[name]
    adj, feats, gi, y = b.to_torch(dev)
    st = distance_bias_structure(adj, gi, max_distance=max_distance)
    a = dense_adj(b)
    _, dists = oracle.apply_batch(a, b.sizes, np.zeros((b.n_nodes, 1)), max_distance=max_distance)
    start = st.row_start.cpu().numpy()
    col = st.col.cpu().numpy()
    dist = st.dist.cpu().numpy()
    levels = st.level_count.cpu().numpy()
    lo = 0
    for ng, dm in zip(b.sizes, dists):
        for i in range(ng):
            want = [(int(dm[i, j]), lo + j) for j in range(ng) if 1 <= dm[i, j] <= max_distance]
            want.sort()
            s, e = start[lo + i], start[lo + i + 1]
            got = list(zip(dist[s:e].tolist(), col[s:e].tolist()))
            assert got == want, (name, lo + i)
            for d in range(1, max_distance + 1):
                assert levels[lo + i, d - 1] == sum(1 for w in want if w[0] == d)
        lo += int(ng)
    assert start[-1] == col.size


@pytest.mark.parametrize("name", ["sparse", "denser", "hubs", "one_graph", "tiny"])
@pytest.mark.parametrize("walk_type,alpha,c,max_distance", [("exp", 1.0, 0.1, 3), ("levy", 1.5, 0.3, 4),
                                                            ("exp", 0.5, 0.0, 2)])
def test_operator_forward_backward(dev, name, walk_type, alpha, c, max_distance):
    """P x and P^T g against the dense float64 matrices; fp32 sums over up to 300 terms: rtol 1e-5 of the largest
    magnitude."""
    from gnn_amd.distance_bias import distance_bias_propagate, distance_bias_structure
    b = batches()[name]
    adj, feats, gi, y = b.to_torch(dev)
    rng = np.random.default_rng(7)
    d = 5
    x = rng.normal(size=(b.n_nodes, d)).astype(np.float32)
    g = rng.normal(size=(b.n_nodes, d)).astype(np.float32)
    a = dense_adj(b)
    want, dists = oracle.apply_batch(a, b.sizes, x, max_distance, walk_type, alpha, c)
    want_t = np.zeros_like(want)
    lo = 0
    for ng, dm in zip(b.sizes, dists):
        p = oracle.walk_probabilities(dm, walk_type, alpha, c)
        want_t[lo:lo + ng] = p.T @ g[lo:lo + ng].astype(np.float64)
        lo += int(ng)
    st = distance_bias_structure(adj, gi, max_distance, walk_type, alpha, c)
    xd = torch.from_numpy(x).to(dev).requires_grad_(True)
    out = distance_bias_propagate(xd, st)
    out.backward(torch.from_numpy(g).to(dev))
    got, got_t = out.detach().cpu().numpy(), xd.grad.cpu().numpy()
    assert np.abs(got - want).max() <= 1e-5 * max(1.0, np.abs(want).max())
    assert np.abs(got_t - want_t).max() <= 1e-5 * max(1.0, np.abs(want_t).max())
    # rows of P sum to one (graphs of more than one node): P 1 = 1
    ones = torch.ones(b.n_nodes, 3, device=dev)
    p1 = distance_bias_propagate(ones, st).cpu().numpy()
    big = np.repeat(b.sizes > 1, b.sizes)
    assert np.abs(p1[big] - 1.0).max() < 1e-5
    # bitwise reproducible
    assert torch.equal(distance_bias_propagate(xd.detach(), st), out.detach())


def test_distance_bias_errors(dev):
    from gnn_amd.distance_bias import distance_bias_propagate, distance_bias_structure
    from gnn_amd.synth import random_batch
    b = random_batch(3, 5, 9, 2.0, seed=9)
    adj, feats, gi, y = b.to_torch(dev)
    with pytest.raises(ValueError):
        distance_bias_structure(adj, gi, max_distance=0)
    with pytest.raises(ValueError):
        distance_bias_structure(adj, gi, walk_type="other")
    st = distance_bias_structure(adj, gi)
    with pytest.raises(ValueError):
        distance_bias_propagate(torch.zeros(b.n_nodes + 1, 4, device=dev), st)
    with pytest.raises(RuntimeError):
        distance_bias_propagate(torch.zeros(b.n_nodes, 4), st)


def test_full_size_properties(dev):
    """The headline batch (1024 graphs, 307 842 nodes, 17.6 M pairs within three hops), through properties that
    need no dense matrix: rows of P sum to one; <P x, g> = <x, P^T g> (forward and backward are adjoint); stored
    pairs are symmetric in count per distance (hop distance is symmetric); the one-hop level is the adjacency
    matrix itself."""
    from gnn_amd.distance_bias import distance_bias_propagate, distance_bias_structure
    from gnn_amd.synth import config_batch
    b = config_batch("proteins1024", 1)
    adj, feats, gi, y = b.to_torch(dev)
    st = distance_bias_structure(adj, gi, max_distance=3, walk_type="exp", alpha=1.0, c=0.1)
    n = b.n_nodes
    ones = torch.ones(n, 2, device=dev)
    assert float((distance_bias_propagate(ones, st) - 1.0).abs().max()) < 1e-5
    gen = torch.Generator(device="cpu").manual_seed(3)
    x = torch.randn(n, 4, generator=gen).to(dev).requires_grad_(True)
    g = torch.randn(n, 4, generator=gen).to(dev)
    out = distance_bias_propagate(x, st)
    out.backward(g)
    lhs = float((out.detach().double() * g.double()).sum())
    rhs = float((x.detach().double() * x.grad.double()).sum())
    assert abs(lhs - rhs) <= 1e-5 * max(abs(lhs), abs(rhs), 1.0)
    # integer structure: level one is A (same row pointers and columns), and each distance occurs equally often
    # as (i, j) and (j, i)
    lc = st.level_count.long()
    assert torch.equal(lc[:, 0], (st.base.csr.crow[1:] - st.base.csr.crow[:-1]).long())
    rows = torch.repeat_interleave(torch.arange(n, device=dev), (st.row_start[1:] - st.row_start[:-1]).long())
    for d in (1, 2, 3):
        sel = st.dist == d
        by_row = torch.bincount(rows[sel], minlength=n)
        by_col = torch.bincount(st.col[sel].long(), minlength=n)
        assert torch.equal(by_row, by_col)
        assert torch.equal(by_row, lc[:, d - 1])


@pytest.mark.parametrize("name", ["cora", "pubmed"])
def test_planetoid_scale_single_graph(dev, name):
    """The operator on the Cora- and PubMed-shaped graphs (2708 / 19 717 nodes, 10 556 / 88 648 stored entries, one
    graph): rows of P x for a sample of source nodes against P rows built from scipy's hop-limited shortest paths
    (the reference's other route to the distance matrix, rwgnn2/rwgnn_dis_bias6.py:130-154) in float64 - the dense
    all-pairs restatement is 3 GB at PubMed's size -, plus the properties that need no matrix."""
    from scipy.sparse import csr_matrix
    from scipy.sparse.csgraph import dijkstra
    from gnn_amd.distance_bias import distance_bias_propagate, distance_bias_structure
    from gnn_amd.synth import config_batch
    b = config_batch(name, 1)
    adj, feats, gi, y = b.to_torch(dev)
    n = b.n_nodes
    max_distance, alpha, c = 3, 1.0, 0.1
    st = distance_bias_structure(adj, gi, max_distance=max_distance, walk_type="exp", alpha=alpha, c=c)
    gen = torch.Generator(device="cpu").manual_seed(11)
    x = torch.randn(n, 8, generator=gen).to(dev).requires_grad_(True)
    g = torch.randn(n, 8, generator=gen).to(dev)
    out = distance_bias_propagate(x, st)
    out.backward(g)
    a = csr_matrix((b.values, (b.indices[0], b.indices[1])), shape=(n, n))
    sample = np.random.default_rng(2).choice(n, size=64, replace=False)
    dist = dijkstra(a, directed=False, indices=sample, unweighted=True, limit=max_distance)
    dist[np.isinf(dist)] = max_distance + 1
    w = np.exp(-alpha * dist)
    w[np.arange(sample.size), sample] = 0.0
    rs = w.sum(axis=1, keepdims=True)
    rs[rs == 0] = 1
    x64 = x.detach().cpu().numpy().astype(np.float64)
    want = (1 - c) * (w / rs) @ x64 + c / n * x64.sum(axis=0, keepdims=True)
    got = out.detach().cpu().numpy()[sample]
    assert np.abs(got - want).max() <= 1e-5 * max(1.0, np.abs(want).max())
    ones = torch.ones(n, 2, device=dev)
    assert float((distance_bias_propagate(ones, st) - 1.0).abs().max()) < 1e-5
    lhs = float((out.detach().double() * g.double()).sum())
    rhs = float((x.detach().double() * x.grad.double()).sum())
    assert abs(lhs - rhs) <= 1e-5 * max(abs(lhs), abs(rhs), 1.0)
    assert torch.equal(st.level_count.long()[:, 0], (st.base.csr.crow[1:] - st.base.csr.crow[:-1]).long())
