"""The propagation chain per graph in LDS for block-diagonal batches of large graphs (csrc/rwk_chain.hip):
X_i = A X_{i-1} (rwgnn2/model.py:55) with the projection (model.py:43) in front, and the backward
dX_{i-1} += A^T dX_i with the projection's backward behind, against float64 evaluations of the same products and
against the SpMM launches they replace (same bits for rows of at most 16 stored entries; longer rows are summed by
several lanes, in another fixed order)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL = 1e-5


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def batches():
    from gnn_amd.synth import powerlaw_batch, random_batch
    return {
        "hubs_f1": powerlaw_batch(7, 300, 2500, 2.1, seed=3),                 # hubs of hundreds of entries
        "hubs_f3": powerlaw_batch(5, 900, 1800, 2.1, seed=4, n_feat=3),
        "big_sparse_f1": random_batch(4, 1100, 2600, 3.7, seed=5),            # no hubs, 1024 threads per graph
        "big_sparse_f16": random_batch(3, 1100, 1500, 5.0, seed=6, n_feat=16),
        "small_hubs": powerlaw_batch(9, 80, 200, 1.8, seed=7),                # 256 threads per graph, rows above 64
        "max_lds": random_batch(2, 3990, 4000, 3.0, seed=8),                  # the largest graphs that fit
        # ~2000 rows of ~40 entries per graph: more 16-lane rows than the list holds (the rest fall back to their own
        # thread) and more stored entries than the LDS share of the columns (the rest are read from memory)
        "dense_overflow": random_batch(2, 2000, 2100, 40.0, seed=9),
    }


def dense64(adj):
    a = adj.coalesce()
    return torch.sparse_coo_tensor(a.indices(), a.values().double(), a.shape).coalesce()


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("name", list(batches()))
@pytest.mark.parametrize("steps", [2, 4])
def test_chain_forward_kernel(name, steps, dev):
    from gnn_amd import _lib
    from gnn_amd.eval import _facts
    from gnn_amd.structure import get_structure
    lib = _lib.load()
    b = batches()[name]
    adj, feats, gi, _ = b.to_torch(dev)
    st = get_structure(adj, gi)
    max_nodes, block_diagonal = _facts(st)
    assert block_diagonal
    n, f = feats.shape
    gen = torch.Generator(device="cpu").manual_seed(11)
    w = torch.randn(4, f, generator=gen).to(dev) * 0.3
    bias = torch.randn(4, generator=gen).to(dev) * 0.1
    h, s = 3, 5
    adj_h = torch.rand(h, s * (s - 1) // 2, generator=gen).to(dev) - 0.3
    feat_h = torch.rand(h, s, 4, generator=gen).to(dev)
    assert lib.rwk_graph_chain_forward_lds(f, max_nodes, s) > 0
    stream = _lib.current_stream(dev)

    def run():
        xs = torch.full((steps, n, 4), float("nan"), device=dev)
        ah = torch.empty(h, s, s, device=dev)
        z = torch.empty(steps, h, s, 4, device=dev)
        csr = st.csr
        _lib.check(lib.rwk_graph_chain_forward(
            feats.data_ptr(), w.data_ptr(), bias.data_ptr(), n, f, csr.crow.data_ptr(), csr.col.data_ptr(),
            None if csr.unit_values else csr.val.data_ptr(), st.seg_ptr.data_ptr(), st.n_graphs, max_nodes, steps,
            xs.data_ptr(), adj_h.data_ptr(), feat_h.data_ptr(), h, s, ah.data_ptr(), z.data_ptr(), None, None,
            stream), "rwk_graph_chain_forward")
        return xs, ah, z

    xs, ah, z = run()
    # float64
    a64 = dense64(adj)
    x = torch.sigmoid(feats.double() @ w.double().T + bias.double())
    for i in range(steps):
        assert rel(xs[i], x) < TOL, i
        x = torch.sparse.mm(a64, x)
    # the launches it replaces: projection + hidden graphs, then one SpMM per step
    xs2 = torch.empty_like(xs)
    ah2, z2 = torch.empty_like(ah), torch.empty_like(z)
    _lib.check(lib.rwk_project_hidden_forward(feats.data_ptr(), w.data_ptr(), bias.data_ptr(), n, f, 4, xs2.data_ptr(),
                                              adj_h.data_ptr(), feat_h.data_ptr(), h, s, steps, ah2.data_ptr(),
                                              z2.data_ptr(), None, None, stream), "rwk_project_hidden_forward")
    for i in range(1, steps):
        st.spmm(xs2[i - 1], xs2[i])
    assert torch.equal(ah, ah2) and torch.equal(z, z2)
    assert torch.equal(xs[0], xs2[0])
    if st.csr.max_degree <= 16:
        assert torch.equal(xs, xs2), "rows of at most 16 entries: the SpMM's own order of additions"
    else:
        deg = (st.csr.crow[1:] - st.csr.crow[:-1])
        short = deg <= 16
        assert torch.equal(xs[1][short], xs2[1][short])
        assert rel(xs, xs2) < 1e-6
    xs3, _, _ = run()
    assert torch.equal(xs, xs3), "bitwise reproducible"


@pytest.mark.parametrize("name", list(batches()))
@pytest.mark.parametrize("steps,with_d_feat,job_rows", [(2, False, 0), (4, True, 0), (3, False, 77)])
def test_chain_backward_kernel(name, steps, with_d_feat, job_rows, dev):
    from gnn_amd import _lib
    from gnn_amd.eval import _facts
    from gnn_amd.structure import get_structure
    lib = _lib.load()
    b = batches()[name]
    adj, feats, gi, _ = b.to_torch(dev)
    st = get_structure(adj, gi)
    max_nodes, _ = _facts(st)
    n, f = feats.shape
    gen = torch.Generator(device="cpu").manual_seed(12)
    w = torch.randn(4, f, generator=gen).to(dev) * 0.3
    x0 = torch.rand(n, 4, generator=gen).to(dev)
    dx = torch.randn(steps, n, 4, generator=gen).to(dev)
    assert lib.rwk_graph_chain_backward_lds(f, max_nodes) > 0
    stream = _lib.current_stream(dev)
    csr_t = st.csr_t
    red_in = torch.randn(job_rows, 130, generator=gen).to(dev) if job_rows else None
    red_out = torch.full((32, 130), float("nan"), device=dev) if job_rows else None

    def run():
        partials = torch.full((st.n_graphs, 4 * f + 4), float("nan"), device=dev)
        d_feat = torch.full((n, f), float("nan"), device=dev) if with_d_feat else None
        keep = dx.clone()
        _lib.check(lib.rwk_graph_chain_backward(
            dx.data_ptr(), x0.data_ptr(), feats.data_ptr(), w.data_ptr(), n, f, csr_t.crow.data_ptr(),
            csr_t.col.data_ptr(), None if csr_t.unit_values else csr_t.val.data_ptr(), st.seg_ptr.data_ptr(),
            st.n_graphs, max_nodes, steps, partials.data_ptr(), _lib.ptr(d_feat), _lib.ptr(red_in), job_rows,
            130 if job_rows else 0, _lib.ptr(red_out), stream), "rwk_graph_chain_backward")
        assert torch.equal(dx, keep), "dx is only read"
        return partials, d_feat

    partials, d_feat = run()
    at64 = dense64(adj).t().coalesce()
    g = dx[steps - 1].double()
    for i in range(steps - 1, 0, -1):
        g = dx[i - 1].double() + torch.sparse.mm(at64, g)
    dpre = g * x0.double() * (1 - x0.double())
    want = torch.cat([(dpre.T @ feats.double()).reshape(-1), dpre.sum(0)])
    got = partials.double().sum(0)
    floor = 6e-8 * np.sqrt(n) * 4 * float(dpre.abs().max() * feats.abs().max())
    assert float((got - want).abs().max()) <= TOL * float(want.abs().max()) + floor
    # per graph
    seg = st.seg_ptr.cpu().numpy()
    for gidx in (0, st.n_graphs - 1):
        lo, hi = int(seg[gidx]), int(seg[gidx + 1])
        w_g = torch.cat([(dpre[lo:hi].T @ feats[lo:hi].double()).reshape(-1), dpre[lo:hi].sum(0)])
        assert float((partials[gidx].double() - w_g).abs().max()) <= TOL * float(w_g.abs().max()) + floor
    if with_d_feat:
        assert rel(d_feat, dpre @ w.double()) < TOL
    if job_rows:
        assert rel(red_out.sum(0), red_in.double().sum(0)) < TOL
    p2, f2 = run()
    assert torch.equal(partials, p2), "bitwise reproducible"
    if with_d_feat:
        assert torch.equal(d_feat, f2)


@pytest.mark.parametrize("name,steps", [("hubs_f1", 4), ("hubs_f3", 3), ("big_sparse_f1", 2), ("small_hubs", 4),
                                        ("dense_overflow", 3)])
@pytest.mark.parametrize("train", [False, True])
def test_layer_with_chain_against_spmm_launches(name, steps, train, dev, monkeypatch):
    """The layer (forward, backward) with the chain launches and with the SpMM launches they replace: K, every
    gradient and the generator state.  Without hubs K and the hidden-graph gradients are the same bits."""
    import gnn_amd.function as fn
    from gnn_amd import RW_NN
    from gnn_amd.structure import get_structure
    monkeypatch.setattr(fn, "GRAM_FAST_PATH", False)
    monkeypatch.setattr(fn, "GRAPH_CHAIN_MIN_GRAPHS", 1)      # (by default only batches that give every CU a graph)
    b = batches()[name]
    adj, feats, gi, _ = b.to_torch(dev)
    st = get_structure(adj, gi)
    gen = torch.Generator(device="cpu").manual_seed(13)
    dk = torch.randn(b.n_graphs, steps * 16, generator=gen).to(dev)
    runs = []
    for chain in (True, False):
        monkeypatch.setattr(fn, "GRAPH_CHAIN", chain)
        torch.manual_seed(3)
        m = RW_NN(feats.shape[1], steps, 16, 5, 4, 32, False, 2, 0.3 if train else 0.0, dev).to(dev)
        m.train(train)
        if train:
            m.seed_dropout(7)
        fe = feats.clone().requires_grad_(True)
        k = m.rw_features(adj, fe, gi)
        k.backward(dk)
        runs.append({"k": k.detach(), "adj_hidden": m.adj_hidden.grad, "features_hidden": m.features_hidden.grad,
                     "fc.weight": m.fc.weight.grad, "fc.bias": m.fc.bias.grad, "features": fe.grad,
                     "rng": m._rng_state.clone()})
    a, c = runs
    assert torch.equal(a["rng"], c["rng"])
    hubs = max(st.csr.max_degree, st.csr_t.max_degree) > 16
    for key in ("k", "adj_hidden", "features_hidden", "features"):
        if hubs:
            assert rel(a[key], c[key]) < 2e-6, key
        else:
            assert torch.equal(a[key], c[key]), key
    floor = 6e-8 * np.sqrt(b.n_nodes) * 4 * float(a["features"].abs().max()) * float(feats.abs().max() + 1)
    for key in ("fc.weight", "fc.bias"):          # the same per-node terms added per graph instead of per 256 rows
        assert float((a[key] - c[key]).abs().max()) <= TOL * float(c[key].abs().max()) + floor, key
