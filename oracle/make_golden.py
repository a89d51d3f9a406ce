"""Generate ``tests/golden/*.npz`` by RUNNING THE REFERENCE's own ``RW_NN``.

Run only in the build container (``/root/reference`` does not exist on the GPU box)::

    python oracle/make_golden.py

It imports ``model.RW_NN`` and ``utils.load_data/generate_batches`` from
``/root/reference/rwgnn2`` unchanged, feeds them seeded inputs, and stores inputs,
parameters and every output the parity tests compare against:

* ``coo_indices`` / ``coo_values`` / ``features`` / ``graph_indicator`` / ``y``
* ``csr_crow`` / ``csr_col`` / ``csr_val``   (``adj.coalesce().to_sparse_csr()``)
* ``counts``                                 (``torch.unique(..., return_counts=True)``)
* ``param.<name>`` for every ``state_dict`` entry
* ``k``    -- the (G, L*H) input of ``bn`` (forward pre-hook), i.e. model.py:67
* ``dk``   -- the gradient of the loss with respect to ``k`` (tensor hook on the same input)
* ``out``  -- log-probabilities, ``loss`` -- ``F.cross_entropy(out, y)``
* ``grad.<name>`` for every parameter and ``grad.features``

Only data is written; no reference source text is copied.
"""
from __future__ import annotations

import os
import sys
import warnings

import numpy as np

warnings.filterwarnings("ignore")
REF = "/root/reference/rwgnn2"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def _import_reference():
    sys.path.insert(0, REF)
    import torch  # noqa: F401
    from model import RW_NN  # type: ignore
    from utils import generate_batches, load_data  # type: ignore
    return RW_NN, load_data, generate_batches


def run_case(name, RW_NN, adj, features, graph_indicator, y, hp, seed, train=True, bn_stats_seed=None):
    import torch
    import torch.nn.functional as F

    torch.manual_seed(seed)
    model = RW_NN(hp["input_dim"], hp["max_step"], hp["hidden_graphs"], hp["size_hidden_graphs"],
                  hp["hidden_dim"], hp["penultimate_dim"], hp["normalize"], hp["n_classes"],
                  hp["dropout"], "cpu")
    if bn_stats_seed is not None:
        g = torch.Generator().manual_seed(bn_stats_seed)
        model.bn.running_mean.copy_(torch.rand(model.bn.running_mean.shape, generator=g))
        model.bn.running_var.copy_(torch.rand(model.bn.running_var.shape, generator=g) + 0.5)
    model.train(train)
    state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    captured = {}

    def _grab(mod, inp):
        captured["k"] = inp[0].detach().clone()
        inp[0].register_hook(lambda g: captured.__setitem__("dk", g.detach().clone()))

    model.bn.register_forward_pre_hook(_grab)
    features = features.clone().requires_grad_(True)
    out = model(adj, features, graph_indicator)
    loss = F.cross_entropy(out, y)
    loss.backward()
    coal = adj.coalesce()
    csr = coal.to_sparse_csr()
    _, counts = torch.unique(graph_indicator, return_counts=True)
    data = {
        "coo_indices": adj._indices().numpy(),
        "coo_values": adj._values().numpy(),
        "n_nodes": np.int64(adj.shape[0]),
        "features": features.detach().numpy(),
        "graph_indicator": graph_indicator.numpy(),
        "y": y.numpy(),
        "csr_crow": csr.crow_indices().numpy(),
        "csr_col": csr.col_indices().numpy(),
        "csr_val": csr.values().numpy(),
        "counts": counts.numpy(),
        "k": captured["k"].numpy(),
        "dk": captured["dk"].numpy(),
        "out": out.detach().numpy(),
        "loss": loss.detach().numpy(),
        "train": np.bool_(train),
        "grad.features": features.grad.numpy(),
    }
    for key, val in hp.items():
        data["hp." + key] = np.asarray(val)
    for key, val in state.items():
        data["param." + key] = val.numpy()
    for key, p in model.named_parameters():
        if p.grad is not None:            # max_step == 1 never touches adj_hidden: grad stays None
            data["grad." + key] = p.grad.numpy()
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **data)
    print(f"{name}: N={adj.shape[0]} nnz_in={adj._nnz()} nnz_csr={csr.values().numel()} "
          f"G={counts.numel()} loss={float(loss):.6f}")


def random_graph_batch(rng, sizes, p_edge, directed, weighted, n_feat, dup_frac=0.0, shuffle_entries=False,
                       permute_nodes=False):
    """A block-diagonal random batch as legacy-constructed sparse COO (possibly uncoalesced)."""
    import torch
    rows, cols, vals, gi = [], [], [], []
    base = 0
    for g, n in enumerate(sizes):
        m = rng.random((n, n)) < p_edge
        np.fill_diagonal(m, False)
        if not directed:
            m = np.triu(m, 1)
            m = m | m.T
        r, c = np.nonzero(m)
        if weighted:
            v = rng.uniform(0.25, 2.0, size=r.size).astype(np.float32)
            if not directed:
                w = np.zeros((n, n), np.float32)
                w[r, c] = v
                w = np.triu(w, 1)
                w = w + w.T
                v = w[r, c]
        else:
            v = np.ones(r.size, np.float32)
        rows.append(r + base)
        cols.append(c + base)
        vals.append(v)
        gi.append(np.full(n, g, np.int64))
        base += n
    rows = np.concatenate(rows)
    cols = np.concatenate(cols)
    vals = np.concatenate(vals)
    gi = np.concatenate(gi)
    n_total = base
    if dup_frac > 0 and rows.size:
        k = max(1, int(dup_frac * rows.size))
        pick = rng.integers(0, rows.size, size=k)
        rows = np.concatenate([rows, rows[pick]])
        cols = np.concatenate([cols, cols[pick]])
        vals = np.concatenate([vals, rng.uniform(0.1, 1.0, size=k).astype(np.float32)])
    if shuffle_entries:
        order = rng.permutation(rows.size)
        rows, cols, vals = rows[order], cols[order], vals[order]
    feats = rng.uniform(-1.0, 1.0, size=(n_total, n_feat)).astype(np.float32)
    if permute_nodes:
        perm = rng.permutation(n_total)           # new position of old node
        inv = np.empty_like(perm)
        inv[perm] = np.arange(n_total)
        rows, cols = perm[rows], perm[cols]
        gi = gi[inv]
        feats = feats[inv]
    idx = torch.from_numpy(np.vstack([rows, cols]).astype(np.int64))
    adj = torch.sparse_coo_tensor(idx, torch.from_numpy(vals), (n_total, n_total))
    return adj, torch.from_numpy(feats), torch.from_numpy(gi)


def main():
    import torch

    RW_NN, load_data, generate_batches = _import_reference()
    cwd = os.getcwd()
    os.chdir(REF)                      # load_data uses cwd-relative "datasets/..."
    try:
        adj_l, feat_l, labels = load_data("MUTAG", False)
    finally:
        os.chdir(cwd)
    labels = (labels > 0).astype(np.int64)
    y_l = [np.array(v) for v in labels]
    adj_b, feat_b, gi_b, y_b = generate_batches(adj_l, feat_l, y_l, 64, "cpu")

    default = dict(input_dim=1, max_step=2, hidden_graphs=16, size_hidden_graphs=5, hidden_dim=4,
                   penultimate_dim=32, normalize=False, n_classes=2, dropout=0.0)
    run_case("mutag_b0", RW_NN, adj_b[0], feat_b[0], gi_b[0], y_b[0], default, seed=0)
    run_case("mutag_b2_norm", RW_NN, adj_b[2], feat_b[2], gi_b[2], y_b[2], dict(default, normalize=True), seed=1)
    run_case("mutag_b1_eval", RW_NN, adj_b[1], feat_b[1], gi_b[1], y_b[1], dict(default, dropout=0.2), seed=2,
             train=False, bn_stats_seed=7)

    rng = np.random.default_rng(20240611)

    adj, f, gi = random_graph_batch(rng, [5, 7, 3], 0.4, directed=True, weighted=True, n_feat=3,
                                    dup_frac=0.3, shuffle_entries=True)
    y = torch.tensor([0, 2, 1])
    run_case("directed3_dups", RW_NN, adj, f, gi, y,
             dict(input_dim=3, max_step=3, hidden_graphs=3, size_hidden_graphs=4, hidden_dim=6,
                  penultimate_dim=5, normalize=True, n_classes=3, dropout=0.0), seed=3)

    adj, f, gi = random_graph_batch(rng, [9, 4, 12, 6], 0.3, directed=False, weighted=False, n_feat=2,
                                    permute_nodes=True)
    y = torch.tensor([1, 0, 0, 1])
    run_case("unsorted_indicator", RW_NN, adj, f, gi, y,
             dict(input_dim=2, max_step=2, hidden_graphs=5, size_hidden_graphs=3, hidden_dim=8,
                  penultimate_dim=7, normalize=False, n_classes=2, dropout=0.0), seed=4)

    adj, f, gi = random_graph_batch(rng, [150], 0.03, directed=False, weighted=False, n_feat=20)
    y = torch.tensor([4])
    # G = 1: batch-norm in train mode needs more than one row, so this case runs in eval mode.
    run_case("single_graph_eval", RW_NN, adj, f, gi, y,
             dict(input_dim=20, max_step=3, hidden_graphs=8, size_hidden_graphs=4, hidden_dim=16,
                  penultimate_dim=12, normalize=True, n_classes=7, dropout=0.5), seed=5,
             train=False, bn_stats_seed=11)

    sizes = [1, 2, 1, 3, 1, 1, 2, 70, 1, 2]
    adj, f, gi = random_graph_batch(rng, sizes, 0.6, directed=False, weighted=True, n_feat=1)
    y = torch.tensor([0, 1, 0, 1, 1, 0, 0, 1, 1, 0])
    run_case("tiny_graphs_step4", RW_NN, adj, f, gi, y,
             dict(input_dim=1, max_step=4, hidden_graphs=2, size_hidden_graphs=2, hidden_dim=4,
                  penultimate_dim=4, normalize=False, n_classes=2, dropout=0.0), seed=6)

    adj, f, gi = random_graph_batch(rng, [12, 30, 17, 25, 10, 40], 0.12, directed=False, weighted=False, n_feat=16)
    y = torch.tensor([0, 1, 1, 0, 1, 0])
    run_case("wide_h64_s10_d64", RW_NN, adj, f, gi, y,
             dict(input_dim=16, max_step=2, hidden_graphs=64, size_hidden_graphs=10, hidden_dim=64,
                  penultimate_dim=32, normalize=False, n_classes=2, dropout=0.0), seed=7)

    adj, f, gi = random_graph_batch(rng, [20, 33, 8], 0.2, directed=False, weighted=False, n_feat=4)
    y = torch.tensor([1, 1, 0])
    run_case("step1", RW_NN, adj, f, gi, y,
             dict(input_dim=4, max_step=1, hidden_graphs=16, size_hidden_graphs=5, hidden_dim=4,
                  penultimate_dim=8, normalize=True, n_classes=2, dropout=0.0), seed=8)

    adj, f, gi = random_graph_batch(rng, [300, 280, 350], 0.012, directed=False, weighted=False, n_feat=1)
    y = torch.tensor([1, 0, 0])
    run_case("powerlawish_d32_step4", RW_NN, adj, f, gi, y,
             dict(input_dim=1, max_step=4, hidden_graphs=16, size_hidden_graphs=5, hidden_dim=32,
                  penultimate_dim=16, normalize=True, n_classes=2, dropout=0.0), seed=9)


if __name__ == "__main__":
    main()
