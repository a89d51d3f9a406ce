"""Parity of the HIP path (through the C ABI) against the CPU oracle and the reference's golden vectors.

Tolerances (float32 kernels vs a float64 restatement): max|a-b| <= TOL * max|b| per tensor with
TOL = 1e-5, the bound BASELINE.json's north_star names.  Integer structure is compared bit for bit.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from _golden import GoldenCase, case_names, rel_err
from oracle import rwnn_oracle as O

pytestmark = pytest.mark.gpu

TOL = 1e-5
CASES = case_names()


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _model_for(c, dev, train=None):
    from gnn_amd import RW_NN
    hp = c.hp
    m = RW_NN(hp["input_dim"], hp["max_step"], hp["hidden_graphs"], hp["size_hidden_graphs"], hp["hidden_dim"],
              hp["penultimate_dim"], bool(hp["normalize"]), hp["n_classes"], hp["dropout"], dev)
    missing = m.load_state_dict(c.torch_state(), strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    m = m.to(dev)
    m.train(c.train if train is None else train)
    return m


def _x0(c):
    w = c.params["fc.weight"].astype(np.float64)
    b = c.params["fc.bias"].astype(np.float64)
    return 1.0 / (1.0 + np.exp(-(c["features"].astype(np.float64) @ w.T + b)))


@pytest.mark.parametrize("name", CASES)
def test_structure_bit_exact(name, dev):
    from gnn_amd import GraphStructure
    c = GoldenCase(name)
    adj, _, gi, _ = c.torch_inputs(dev)
    st = GraphStructure.build(adj, gi)
    assert np.array_equal(st.csr.crow.cpu().numpy().astype(np.int64), c["csr_crow"])
    assert np.array_equal(st.csr.col.cpu().numpy().astype(np.int64), c["csr_col"])
    assert np.array_equal(st.csr.val.cpu().numpy().view(np.uint32), c["csr_val"].view(np.uint32))
    assert st.n_graphs == c["counts"].size
    assert np.array_equal(st.counts.cpu().numpy(), c["counts"])
    # graph id per sorted position and the permutation reproduce graph_indicator
    gid = st.gid.cpu().numpy().astype(np.int64)
    assert np.all(np.diff(gid) >= 0)
    if st.perm is None:
        assert np.array_equal(gid, c["graph_indicator"])
    else:
        assert np.array_equal(gid, c["graph_indicator"][st.perm.cpu().numpy()])
    seg = st.seg_ptr.cpu().numpy()
    assert np.array_equal(np.diff(seg).astype(np.int64), c["counts"])
    # transpose CSR equals the oracle's
    t_crow, t_col, t_val = O.csr_transpose(c["csr_crow"], c["csr_col"], c["csr_val"], c.n_nodes)
    assert np.array_equal(st.csr_t.crow.cpu().numpy().astype(np.int64), t_crow)
    assert np.array_equal(st.csr_t.col.cpu().numpy().astype(np.int64), t_col)
    assert np.array_equal(st.csr_t.val.cpu().numpy().view(np.uint32), t_val.view(np.uint32))


@pytest.mark.parametrize("gram", [True, False])
@pytest.mark.parametrize("name", CASES)
def test_layer_forward_backward(name, gram, dev, monkeypatch):
    """K and the layer's VJP (fed the reference's own dL/dK) against the float64 oracle.  The goldens run without
    dropout: at hidden_dim <= 4 that is the Gram-matrix path (gram = True) or the stream kernels with the identity
    mask (gram = False, what every shape ran before the Gram path existed); wider shapes take the stream kernels
    either way."""
    import gnn_amd.function as fn
    monkeypatch.setattr(fn, "GRAM_FAST_PATH", gram)
    c = GoldenCase(name)
    m = _model_for(c, dev)
    adj, feats, gi, _ = c.torch_inputs(dev)
    feats = feats.clone().requires_grad_(True)
    k = m.rw_features(adj, feats, gi)
    csr = (c["csr_crow"], c["csr_col"], c["csr_val"])
    x0 = _x0(c)
    k_ref = O.rw_kernel_forward(csr, x0, c.params["adj_hidden"], c.params["features_hidden"],
                                c["graph_indicator"], c.hp["max_step"], c.hp["normalize"])
    assert k.shape == k_ref.shape
    assert rel_err(k.detach().cpu().numpy(), k_ref) < TOL
    assert rel_err(k.detach().cpu().numpy(), c["k"]) < TOL          # and the reference's own float32 K

    dk = torch.from_numpy(c["dk"]).to(dev)
    k.backward(dk)
    d_x0, d_adj, d_feat = O.rw_kernel_backward(csr, x0, c.params["adj_hidden"], c.params["features_hidden"],
                                               c["graph_indicator"], c.hp["max_step"], c["dk"],
                                               c.hp["normalize"])
    d_pre = d_x0 * x0 * (1 - x0)
    got = {
        "features_hidden": m.features_hidden.grad, "fc.bias": m.fc.bias.grad, "fc.weight": m.fc.weight.grad,
        "features": feats.grad,
    }
    want = {
        "features_hidden": d_feat, "fc.bias": d_pre.sum(0),
        "fc.weight": d_pre.T @ c["features"].astype(np.float64),
        "features": d_pre @ c.params["fc.weight"].astype(np.float64),
    }
    if c.hp["max_step"] > 1:
        got["adj_hidden"] = m.adj_hidden.grad
        want["adj_hidden"] = d_adj
    else:
        assert m.adj_hidden.grad is None
    # fc.bias and fc.weight are sums over all nodes of signed terms d_pre[n] (x features[n]); where they cancel (on
    # mutag_b0 the bias gradient is 3e-4 against a sum of magnitudes of 0.17) no float32 summation order reaches 1e-5
    # of the result, so the rounding floor of a float32 sum of N such terms, eps * sqrt(N) * max|term| (with a factor
    # of 4), is admitted on top - the rule the full-size tests use
    floor = {"fc.bias": np.abs(d_pre).max(), "fc.weight": np.abs(d_pre).max() * np.abs(c["features"]).max()}
    for key in want:
        # Where cancellation over the nodes keeps even the reference's own float32 gradient further than
        # 1e-5 from the float64 truth, we must be at least as accurate as the reference is.
        ref_own_err = rel_err(c.grads[key], want[key])
        err = rel_err(got[key].cpu().numpy(), want[key])
        bound = max(TOL, ref_own_err)
        if key in floor:
            bound += 6e-8 * np.sqrt(c.n_nodes) * 4 * floor[key] / max(np.abs(want[key]).max(), 1e-30)
        assert err < bound, (key, err, bound)


@pytest.mark.parametrize("name", CASES)
def test_layer_element_by_element(name, dev):
    """The same comparison element by element (the checks above are norm-wise): |got - truth| <= 1e-5 |truth| +
    1e-6 max|truth| against the float64 oracle for every element of K; for the gradients the elements outside that
    band are counted and held to twice the count of the reference's own float32 result plus one in a thousand (where
    a sum over the nodes cancels, no float32 summation order keeps every element inside: on the wide case 1.0 % of
    the reference's features_hidden gradient is outside, 1.5 % of ours; every other case and tensor stays within
    the reference's own count plus one in a thousand)."""
    c = GoldenCase(name)
    m = _model_for(c, dev)
    adj, feats, gi, _ = c.torch_inputs(dev)
    feats = feats.clone().requires_grad_(True)
    k = m.rw_features(adj, feats, gi)
    csr = (c["csr_crow"], c["csr_col"], c["csr_val"])
    x0 = _x0(c)
    k_ref = O.rw_kernel_forward(csr, x0, c.params["adj_hidden"], c.params["features_hidden"],
                                c["graph_indicator"], c.hp["max_step"], c.hp["normalize"])

    def outside(got, truth):
        got, truth = np.asarray(got, np.float64), np.asarray(truth, np.float64)
        return int((np.abs(got - truth) > 1e-5 * np.abs(truth) + 1e-6 * np.abs(truth).max()).sum())

    assert outside(k.detach().cpu().numpy(), k_ref) == 0
    k.backward(torch.from_numpy(c["dk"]).to(dev))
    d_x0, d_adj, d_feat = O.rw_kernel_backward(csr, x0, c.params["adj_hidden"], c.params["features_hidden"],
                                               c["graph_indicator"], c.hp["max_step"], c["dk"], c.hp["normalize"])
    d_pre = d_x0 * x0 * (1 - x0)
    pairs = {"features_hidden": (m.features_hidden.grad, d_feat),
             "features": (feats.grad, d_pre @ c.params["fc.weight"].astype(np.float64))}
    if c.hp["max_step"] > 1:
        pairs["adj_hidden"] = (m.adj_hidden.grad, d_adj)
    for key, (got, truth) in pairs.items():
        ours, theirs = outside(got.cpu().numpy(), truth), outside(c.grads[key], truth)
        assert ours <= 2 * theirs + truth.size // 1000, (key, ours, theirs, truth.size)


@pytest.mark.parametrize("name", CASES)
def test_full_model_against_reference(name, dev):
    """Whole forward + cross-entropy backward against the reference's stored outputs.  Gradients behind
    train-mode batch-norm over 3-6 graphs amplify float32 rounding (the reference itself is up to 3.5e-2
    from the float64 truth there, see test_oracle_golden), so those are held to the reference's own
    accuracy; eval-mode cases and all forward outputs are held to 1e-4 / 1e-5."""
    c = GoldenCase(name)
    m = _model_for(c, dev)
    adj, feats, gi, y = c.torch_inputs(dev)
    feats = feats.clone().requires_grad_(True)
    out = m(adj, feats, gi)
    loss = F.cross_entropy(out, y)
    loss.backward()
    assert rel_err(out.detach().cpu().numpy(), c["out"]) < 1e-4
    assert abs(float(loss) - float(c["loss"])) < 1e-4 * max(1.0, abs(float(c["loss"])))
    grads = {n: p.grad for n, p in m.named_parameters()}
    grads["features"] = feats.grad
    truth = _float64_port_grads(c)
    for key, ref in c.grads.items():
        assert grads[key] is not None, key
        ref_own_err = rel_err(ref, truth[key])
        # torch's own train-mode batch-norm differs by 1.2e-4 between its CPU and GPU kernels on the
        # reference's K (x - mean cancels 3-4 digits), before any of our code is involved; the tight
        # checks of the HIP kernels are test_layer_forward_backward and test_dropout_mask_replay
        floor = 5e-3 if c.train else 1e-4
        got = grads[key].cpu().numpy()
        tiny = float(np.max(np.abs(got - truth[key]))) < 2e-6        # e.g. fc2.bias: entries cancel to ~1e-3
        assert tiny or rel_err(got, truth[key]) < max(floor, 10 * ref_own_err), key


def _float64_port_grads(c):
    """Gradients of the whole model in float64 on the CPU (operator-for-operator port of the reference)."""
    from oracle import rwnn_torch_port as P
    adj, feats, gi, y = c.torch_inputs(dtype=torch.float64)
    params = P.cast_params(c.torch_state(), torch.float64, requires_grad=True)
    feats = feats.clone().requires_grad_(True)
    bn_stats = None
    if not c.train:
        bn_stats = (torch.from_numpy(c.params["bn.running_mean"]).double(),
                    torch.from_numpy(c.params["bn.running_var"]).double())
    out, _ = P.forward(params, adj, feats, gi, c.hp["max_step"], c.hp["normalize"], c.hp["dropout"], c.train,
                       bn_stats)
    F.nll_loss(out, y).backward()
    grads = {n: p.grad.numpy() for n, p in params.items() if p.grad is not None}
    grads["features"] = feats.grad.numpy()
    return grads


def _run_layer(c, dev, adj, feats, gi, dk):
    m = _model_for(c, dev)
    fe = feats.clone().requires_grad_(True)
    k = m.rw_features(adj, fe, gi)
    k.backward(dk)
    grads = {n: p.grad for n, p in (("adj_hidden", m.adj_hidden), ("features_hidden", m.features_hidden),
                                    ("fc.weight", m.fc.weight), ("fc.bias", m.fc.bias)) if p.grad is not None}
    return dict(grads, k=k.detach(), features=fe.grad)


@pytest.mark.parametrize("name", ["mutag_b0", "mutag_b2_norm", "step1", "tiny_graphs_step4"])
def test_merged_launches_equal_separate_launches(name, dev):
    """The hidden-graph kernels ride in the projection kernels' launches (forward and backward); with the merge
    switched off the stand-alone kernels run: same bits in K and in every gradient."""
    import gnn_amd.function as fn
    c = GoldenCase(name)
    adj, feats, gi, _ = c.torch_inputs(dev)
    dk = torch.from_numpy(c["dk"]).to(dev)
    runs = []
    try:
        fn.FUSED_STEP_LAUNCHES = False
        for merged in (True, False):
            fn.MERGED_LAUNCHES = merged
            runs.append(_run_layer(c, dev, adj, feats, gi, dk))
    finally:
        fn.MERGED_LAUNCHES = True
        fn.FUSED_STEP_LAUNCHES = True
    assert runs[0].keys() == runs[1].keys()
    for key in runs[0]:
        assert torch.equal(runs[0][key], runs[1][key]), key


@pytest.mark.parametrize("name", ["mutag_b0", "mutag_b1_eval", "mutag_b2_norm", "tiny_graphs_step4",
                                  "unsorted_indicator", "directed3_dups"])
def test_fused_step_launches_equal_separate_launches(name, dev):
    """The first propagation step done inside the projection's launch (forward: X_1 from recomputed neighbour rows;
    backward: A^T dX_1 gathered by the projection's backward pass) against the stand-alone SpMM launches: the same
    additions in the same order, so K and the hidden-graph gradients are the same bits; the projection's weight
    and bias gradients are the same per-node terms added in another (fixed) order."""
    import gnn_amd.function as fn
    c = GoldenCase(name)
    adj, feats, gi, _ = c.torch_inputs(dev)
    dk = torch.from_numpy(c["dk"]).to(dev)
    runs = []
    try:
        for fused in (True, False):
            fn.FUSED_STEP_LAUNCHES = fused
            runs.append(_run_layer(c, dev, adj, feats, gi, dk))
    finally:
        fn.FUSED_STEP_LAUNCHES = True
    assert runs[0].keys() == runs[1].keys()
    for key in ("k", "adj_hidden", "features_hidden"):
        if key in runs[0]:
            assert torch.equal(runs[0][key], runs[1][key]), key
    # per-node gradient behind dX_0 + A^T dX_1, whose two terms partly cancel: the two addition orders differ by a few
    # units in the last place of the larger term (7.7e-7 and 1.1e-6 of the largest entry were measured on
    # mutag_b2_norm with the stream and the Gram backward); each order is held to 1e-5 against the oracle elsewhere
    assert rel_err(runs[0]["features"].cpu().numpy(), runs[1]["features"].cpu().numpy().astype(np.float64)) < 3e-6
    # sums of signed per-node terms: equal up to the rounding of a float32 sum of that many terms
    n_nodes = c.n_nodes
    for key in ("fc.weight", "fc.bias"):
        a, b = runs[0][key].cpu().numpy().astype(np.float64), runs[1][key].cpu().numpy().astype(np.float64)
        scale = np.abs(c["dk"]).max() * max(np.abs(c["features"]).max(), 1.0)
        assert np.abs(a - b).max() <= 1e-5 * np.abs(b).max() + 6e-8 * np.sqrt(n_nodes) * 4 * scale, key


@pytest.mark.parametrize("name", ["mutag_b0", "wide_h64_s10_d64"])
def test_backward_is_bitwise_reproducible(name, dev):
    # the wide case runs the matrix-core kernels with four blocks of hidden graphs: the node-feature gradient of the
    # second and later blocks is added at the L2 by the element's owner lane, in block order
    c = GoldenCase(name)
    m = _model_for(c, dev)
    adj, feats, gi, _ = c.torch_inputs(dev)
    dk = torch.from_numpy(c["dk"]).to(dev)
    runs = []
    for _ in range(3):
        m.zero_grad(set_to_none=True)
        k = m.rw_features(adj, feats, gi)
        k.backward(dk)
        runs.append([k.detach().clone()] + [p.grad.clone() for p in (m.adj_hidden, m.features_hidden, m.fc.weight)])
    for other in runs[1:]:
        for a, b in zip(runs[0], other):
            assert torch.equal(a, b)


def test_keep_bits_from_the_forward_equal_a_second_draw(dev, monkeypatch):
    """Matrix-core path: the backward pass takes the dropout keep bits the forward left in memory; the result is
    the same bits as with the generator run a second time (the other widths always draw again)."""
    from gnn_amd import _lib, get_structure, rw_kernel_layer
    from gnn_amd import function as fn
    c = GoldenCase("wide_h64_s10_d64")
    hp = c.hp
    m = _model_for(c, dev, train=True)
    adj, feats, gi, _ = c.torch_inputs(dev)
    st = get_structure(adj, gi)
    h, s, steps, d = hp["hidden_graphs"], hp["size_hidden_graphs"], hp["max_step"], hp["hidden_dim"]
    words = c.n_nodes * h if s * steps > 32 else (c.n_nodes * h + 1) // 2
    assert _lib.load().rwk_fused_keep_words(c.n_nodes, d, h, s, steps) == words
    dk = torch.from_numpy(c["dk"]).to(dev)
    runs = []
    for on in (True, False):
        monkeypatch.setattr(fn, "KEEP_BITS_HANDOVER", on)
        rng = torch.tensor([0x5EED5EED, 17], dtype=torch.int64, device=dev)
        pre = m.fc(feats).detach().requires_grad_(True)
        adj_h = m.adj_hidden.detach().clone().requires_grad_(True)
        z = m.features_hidden.detach().clone().requires_grad_(True)
        k = rw_kernel_layer(pre, adj_h, z, st, steps, bool(hp["normalize"]), 0.3, rng)
        k.backward(dk)
        runs.append((k.detach().clone(), pre.grad.clone(), adj_h.grad.clone(), z.grad.clone()))
    for a, b in zip(*runs):
        assert torch.equal(a, b)
    assert float(runs[0][1].abs().sum()) > 0


@pytest.mark.parametrize("name,p", [("mutag_b0", 0.2), ("directed3_dups", 0.5), ("powerlawish_d32_step4", 0.1),
                                    ("wide_h64_s10_d64", 0.3), ("unsorted_indicator", 0.25)])
def test_dropout_mask_replay(name, p, dev):
    """Train-mode dropout: dump the kernel's own mask, check it equals the oracle's restatement of the
    generator bit for bit, and replay forward + backward on the CPU with that mask."""
    from gnn_amd import _lib, get_structure, rw_kernel_layer
    c = GoldenCase(name)
    hp = c.hp
    m = _model_for(c, dev, train=True)
    adj, feats, gi, _ = c.torch_inputs(dev)
    st = get_structure(adj, gi)
    seed, offset = 0x1234567890ABCDE, 41
    rng = torch.tensor([seed, offset], dtype=torch.int64, device=dev)
    h, s, steps = hp["hidden_graphs"], hp["size_hidden_graphs"], hp["max_step"]
    keep_dev = torch.empty((steps, h, s, c.n_nodes), dtype=torch.uint8, device=dev)
    lib = _lib.load()
    _lib.check(lib.rwk_dropout_mask(rng.data_ptr(), c.n_nodes, h, s, steps, p, keep_dev.data_ptr(),
                                    _lib.current_stream(dev)), "rwk_dropout_mask")
    keep = keep_dev.cpu().numpy().astype(bool)
    keep_ref = O.dropout_keep(seed, offset, c.n_nodes, h, s, steps, p)
    assert np.array_equal(keep, keep_ref)
    assert abs(keep.mean() - (1 - p)) < 5 * np.sqrt(p * (1 - p) / keep.size) + 2e-5

    pre = m.fc(feats)
    k = rw_kernel_layer(pre, m.adj_hidden, m.features_hidden, st, steps, bool(hp["normalize"]), p, rng)
    assert int(rng[1].item()) == offset + 1            # the launch advanced the device-side offset
    csr = (c["csr_crow"], c["csr_col"], c["csr_val"])
    x0 = _x0(c)
    k_ref = O.rw_kernel_forward(csr, x0, c.params["adj_hidden"], c.params["features_hidden"],
                                c["graph_indicator"], steps, hp["normalize"], keep=keep_ref, drop_p=p)
    assert rel_err(k.detach().cpu().numpy(), k_ref) < TOL
    dk = torch.from_numpy(c["dk"]).to(dev)
    k.backward(dk)
    d_x0, d_adj, d_feat = O.rw_kernel_backward(csr, x0, c.params["adj_hidden"], c.params["features_hidden"],
                                               c["graph_indicator"], steps, c["dk"], hp["normalize"],
                                               keep=keep_ref, drop_p=p)
    d_pre = d_x0 * x0 * (1 - x0)
    assert rel_err(m.features_hidden.grad.cpu().numpy(), d_feat) < TOL
    assert rel_err(m.adj_hidden.grad.cpu().numpy(), d_adj) < TOL
    assert rel_err(m.fc.bias.grad.cpu().numpy(), d_pre.sum(0)) < TOL


def test_spmm_directed_and_long_rows(dev):
    """A x and A^T x on a directed, weighted graph with a few hub rows, through both SpMM schedules: the
    packed kernel (wavefront-per-row pass for the hubs) and the tile kernel (hubs summed by the workgroup,
    one of them spanning several LDS tiles; empty rows; a last workgroup with fewer rows)."""
    from gnn_amd import GraphStructure
    rng = np.random.default_rng(5)
    n = 6001
    rows = [rng.integers(0, n, size=12000), np.full(5000, 7), np.full(400, 1234), rng.integers(0, n, 300)]
    cols = [rng.integers(0, n, size=12000), rng.choice(n, 5000, replace=False), rng.choice(n, 400, replace=False),
            np.full(300, 99)]
    r = np.concatenate(rows)
    cidx = np.concatenate(cols)
    key = np.unique(r * n + cidx)
    r, cidx = key // n, key % n
    v = rng.uniform(-1, 1, size=r.size).astype(np.float32)
    adj = torch.sparse_coo_tensor(torch.from_numpy(np.vstack([r, cidx])), torch.from_numpy(v), (n, n)).to(dev)
    gi = torch.zeros(n, dtype=torch.int64, device=dev)
    st = GraphStructure.build(adj, gi)
    assert not st.symmetric
    assert st.csr.long_rows is not None and st.csr.long_rows.numel() >= 2
    crow, col, val = O.coo_to_csr(np.vstack([r, cidx]), v, n)
    t_crow, t_col, t_val = O.csr_transpose(crow, col, val, n)
    for d in (1, 3, 4, 6, 8, 16, 20, 32, 64):
        x = rng.uniform(-1, 1, size=(n, d)).astype(np.float32)
        xd = torch.from_numpy(x).to(dev)
        ax = O.csr_matmul(crow, col, val, x.astype(np.float64))
        atx = O.csr_matmul(t_crow, t_col, t_val, x.astype(np.float64))
        y0 = torch.from_numpy(rng.uniform(-1, 1, size=(n, d)).astype(np.float32)).to(dev)
        assert st.csr.col16 is not None and not st.csr.unit_values      # 6001 rows: every distance fits 16 bits
        results = {}
        for kernel in ("packed", "tile", "tile32", None):
            y = torch.full_like(xd, float("nan"))
            st.spmm(xd, y, kernel=kernel)
            results[kernel] = y
            assert rel_err(y.cpu().numpy(), ax) < TOL, (d, kernel)
            y1 = y0.clone()
            st.spmm(xd, y1, transpose=True, accumulate=True, kernel=kernel)
            assert rel_err(y1.cpu().numpy(), y0.cpu().numpy().astype(np.float64) + atx) < TOL, (d, kernel)
            y2 = torch.full_like(xd, float("nan"))
            st.spmm(xd, y2, kernel=kernel)
            assert torch.equal(y, y2), "SpMM must be bitwise reproducible"
        assert torch.equal(results["tile"], results["tile32"]), "16-bit column distances must not change a bit"


def test_spmm_compact_structure(dev):
    """The compact structure of the tile kernel: unweighted block-diagonal batches get 16-bit column distances and
    no value array (bit-identical to the int32 / fp32 structure); a column further than 32767 rows away, or a
    weight, switches the respective part off."""
    from gnn_amd import GraphStructure
    rng = np.random.default_rng(11)
    n = 70000

    def build(r, c, v):
        key = np.unique(r.astype(np.int64) * n + c)
        r, c = key // n, key % n
        adj = torch.sparse_coo_tensor(torch.from_numpy(np.vstack([r, c])), torch.from_numpy(v[:r.size].copy()),
                                      (n, n)).to(dev)
        return GraphStructure.build(adj, torch.zeros(n, dtype=torch.int64, device=dev)), (r, c, v[:r.size])

    # near-diagonal, unweighted, directed: columns within +-3000 of the row, including rows near block edges
    r = rng.integers(0, n, size=250000)
    c = np.clip(r + rng.integers(-3000, 3001, size=r.size), 0, n - 1)
    st, (r1, c1, v1) = build(r, c, np.ones(r.size, dtype=np.float32))
    assert st.csr.col16 is not None and st.csr.unit_values and st.csr_t.col16 is not None
    crow, col, val = O.coo_to_csr(np.vstack([r1, c1]), v1, n)
    for d in (4, 6, 16):
        x = rng.uniform(-1, 1, size=(n, d)).astype(np.float32)
        xd = torch.from_numpy(x).to(dev)
        ya, yb = torch.empty_like(xd), torch.empty_like(xd)
        st.spmm(xd, ya, kernel="tile")
        st.spmm(xd, yb, kernel="tile32")
        assert torch.equal(ya, yb)
        assert rel_err(ya.cpu().numpy(), O.csr_matmul(crow, col, val, x.astype(np.float64))) < TOL
        assert st.spmm_bytes_moved(d) == st.nnz * 2 + (n + 1) * 4 + 2 * n * d * 4
    # weights: 16-bit distances stay, the value array is read
    st_w, _ = build(r, c, rng.uniform(0.5, 1.5, size=r.size).astype(np.float32))
    assert st_w.csr.col16 is not None and not st_w.csr.unit_values
    # one far column: back to int32 columns
    st_f, _ = build(np.append(r, 5), np.append(c, 60000), np.ones(r.size + 1, dtype=np.float32))
    assert st_f.csr.col16 is None
    xd = torch.rand((n, 4), device=dev)
    ya, yb = torch.empty_like(xd), torch.empty_like(xd)
    st_f.spmm(xd, ya)
    st_f.spmm(xd, yb, kernel="tile32")
    assert torch.equal(ya, yb)


@pytest.mark.parametrize("weighted", [False, True])
def test_spmm_tile_kernel_on_large_matrices(weighted, dev):
    """The tile SpMM on 2.2 M rows (8600 and 17200 row blocks at one and two float4 chunks per row) of 0..7 entries,
    one row block without any: same bits as the packed kernel's sequential sum per row, for y = A x, y += A x and the
    transposed product; a sample of rows against float64."""
    from gnn_amd import GraphStructure
    rng = np.random.default_rng(23)
    n = 2_200_000
    deg = rng.integers(0, 8, size=n)
    deg[1000:1300] = 0                                      # a whole row block without entries
    rows = np.repeat(np.arange(n), deg)
    off = rng.integers(-2000, 2001, size=rows.size)
    cols = rows + off
    cols = np.where((cols < 0) | (cols >= n), rows - off, cols)      # reflected at the ends: no hub columns
    key = np.unique(rows.astype(np.int64) * n + cols)
    rows, cols = key // n, key % n
    vals = rng.uniform(0.5, 1.5, size=rows.size).astype(np.float32) if weighted else np.ones(rows.size, np.float32)
    adj = torch.sparse_coo_tensor(torch.from_numpy(np.vstack([rows, cols])), torch.from_numpy(vals), (n, n)).to(dev)
    st = GraphStructure.build(adj, torch.zeros(n, dtype=torch.int64, device=dev))
    assert st.csr.col16 is not None and st.csr.unit_values == (not weighted)
    assert max(st.csr.max_degree, st.csr_t.max_degree) <= 64      # (longer rows are summed in another order)
    for d, n_used in ((4, n), (6, n)):
        x = torch.from_numpy(rng.uniform(-1, 1, size=(n, d)).astype(np.float32)).to(dev)
        for transpose in (False, True):
            ya, yb = torch.full_like(x, float("nan")), torch.full_like(x, float("nan"))
            st.spmm(x, ya, transpose=transpose, kernel="tile")
            st.spmm(x, yb, transpose=transpose, kernel="packed")
            assert torch.equal(ya, yb), (d, transpose)
            y0 = torch.from_numpy(rng.uniform(-1, 1, size=(n, d)).astype(np.float32)).to(dev)
            yc, yd = y0.clone(), y0.clone()
            st.spmm(x, yc, transpose=transpose, accumulate=True, kernel="tile")
            st.spmm(x, yd, transpose=transpose, accumulate=True, kernel="packed")
            assert torch.equal(yc, yd), (d, transpose, "accumulate")
            ye = torch.empty_like(x)
            st.spmm(x, ye, transpose=transpose, kernel="tile")
            assert torch.equal(ya, ye), "bitwise reproducible"
        # float64 on the first and last 5000 rows
        for lo, hi in ((0, 5000), (n - 5000, n)):
            sel = (rows >= lo) & (rows < hi)
            want = np.zeros((hi - lo, d))
            np.add.at(want, rows[sel] - lo, vals[sel, None].astype(np.float64) * x.cpu().numpy()[cols[sel]].astype(np.float64))
            y = torch.empty_like(x)
            st.spmm(x, y, kernel="tile")
            assert rel_err(y[lo:hi].cpu().numpy(), want) < TOL


def test_error_behaviour(dev):
    """Inputs the reference rejects are rejected here too (model.py:60 raises on id gaps)."""
    from gnn_amd import GraphStructure
    idx = torch.tensor([[0, 1], [1, 0]], device=dev)
    adj = torch.sparse_coo_tensor(idx, torch.ones(2, device=dev), (4, 4))
    with pytest.raises(IndexError):
        GraphStructure.build(adj, torch.tensor([0, 0, 2, 2], device=dev))
    with pytest.raises(IndexError):
        GraphStructure.build(adj, torch.tensor([1, 1, 2, 2], device=dev))
    with pytest.raises(RuntimeError):
        GraphStructure.build(adj.cpu(), torch.tensor([0, 0, 1, 1]))


@pytest.mark.parametrize("name", ["mutag_b0", "mutag_b2_norm", "step1", "tiny_graphs_step4", "directed3_dups"])
@pytest.mark.parametrize("train_drop", [False, True])
def test_x_chain_launch_equals_separate_launches(name, train_drop, dev, monkeypatch):
    """Projection, every propagation step and the hidden graphs in one launch (one workgroup per graph, neighbours
    gathered from LDS; function.GRAPH_X_CHAIN) against the projection launch followed by one SpMM launch per step:
    K and every gradient bit for bit - without dropout (the goldens' mode) and in training mode with dropout (the
    generator state advanced by the same launch)."""
    import gnn_amd.function as fn
    from gnn_amd.eval import _facts
    from gnn_amd.structure import get_structure
    c = GoldenCase(name)
    adj, feats, gi, _ = c.torch_inputs(dev)
    assert _facts(get_structure(adj, gi))[1], "the golden batches are block-diagonal"
    dk = torch.from_numpy(c["dk"]).to(dev)
    runs = []
    for chain in (True, False):
        monkeypatch.setattr(fn, "GRAPH_X_CHAIN", chain)
        m = _model_for(c, dev, train=train_drop)
        if train_drop:
            m.dropout.p = 0.3
            m.seed_dropout(7)
        fe = feats.clone().requires_grad_(True)
        k = m.rw_features(adj, fe, gi)
        k.backward(dk)
        out = {n: p.grad for n, p in (("adj_hidden", m.adj_hidden), ("features_hidden", m.features_hidden),
                                      ("fc.weight", m.fc.weight), ("fc.bias", m.fc.bias)) if p.grad is not None}
        runs.append(dict(out, k=k.detach(), features=fe.grad, rng=m._rng_state.clone()))
    assert runs[0].keys() == runs[1].keys()
    for key in runs[0]:
        assert torch.equal(runs[0][key], runs[1][key]), key
