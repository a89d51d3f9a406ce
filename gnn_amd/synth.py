"""Seeded synthetic graph batches of the shapes the benchmark configurations name.

Every generator returns host numpy arrays laid out exactly like the output of the reference's
``generate_batches`` (``rwgnn2/utils.py:47-85``): a block-diagonal adjacency as row-major sorted,
duplicate-free COO (int64 indices, float32 ones), float32 features (node degree as the single
feature, ``utils.py:24``, unless ``n_feat`` is given), an int64 ``graph_indicator`` and int64 labels.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

__all__ = ["Batch", "random_batch", "powerlaw_batch", "single_graph_batch", "tu_batch", "config_batch", "CONFIGS"]


@dataclass
class Batch:
    indices: np.ndarray          # (2, nnz) int64, row-major sorted, no duplicates
    values: np.ndarray           # (nnz,) float32
    features: np.ndarray         # (N, F) float32
    graph_indicator: np.ndarray  # (N,) int64
    y: np.ndarray                # (G,) int64
    sizes: np.ndarray            # (G,) nodes per graph

    @property
    def n_nodes(self):
        return int(self.graph_indicator.size)

    @property
    def n_graphs(self):
        return int(self.y.size)

    @property
    def nnz(self):
        return int(self.values.size)

    def slice_graphs(self, lo: int, hi: int) -> "Batch":
        """Graphs [lo, hi) as a batch of their own (node and graph ids renumbered from 0)."""
        starts = np.concatenate([[0], np.cumsum(self.sizes)])
        a, b = int(starts[lo]), int(starts[hi])
        r, c = self.indices
        sel = (r >= a) & (r < b)
        return Batch(np.vstack([r[sel] - a, c[sel] - a]), self.values[sel], self.features[a:b],
                     self.graph_indicator[a:b] - lo, self.y[lo:hi], self.sizes[lo:hi])

    def to_torch(self, device):
        import torch
        n = self.n_nodes
        adj = torch.sparse_coo_tensor(torch.from_numpy(self.indices), torch.from_numpy(self.values), (n, n))
        return (adj.to(device), torch.from_numpy(self.features).to(device),
                torch.from_numpy(self.graph_indicator).to(device), torch.from_numpy(self.y).to(device))


def _finish(rows, cols, sizes, rng, n_feat, n_classes):
    n = int(sizes.sum())
    key = np.unique(rows.astype(np.int64) * n + cols.astype(np.int64))     # sorted row-major, deduplicated
    r, c = key // n, key % n
    vals = np.ones(key.size, np.float32)
    if n_feat is None:
        feats = np.bincount(r, minlength=n).astype(np.float32).reshape(n, 1)
    else:
        feats = rng.uniform(0.0, 1.0, size=(n, n_feat)).astype(np.float32)
    gi = np.repeat(np.arange(sizes.size, dtype=np.int64), sizes)
    y = rng.integers(0, n_classes, size=sizes.size).astype(np.int64)
    return Batch(np.vstack([r, c]), vals, feats, gi, y, sizes.astype(np.int64))


def random_batch(n_graphs, n_lo, n_hi, mean_degree, seed, n_feat=None, n_classes=2) -> Batch:
    """Undirected G(n, m) graphs, n ~ U{n_lo..n_hi}, m = mean_degree * n / 2 edges each."""
    rng = np.random.default_rng(seed)
    sizes = rng.integers(n_lo, n_hi + 1, size=n_graphs)
    rows, cols = [], []
    base = 0
    for n in sizes:
        m = max(1, int(round(mean_degree * n / 2)))
        u = rng.integers(0, n, size=m)
        v = rng.integers(0, n - 1, size=m) if n > 1 else u
        v = np.where(v >= u, v + 1, v) if n > 1 else v     # v != u
        keep = u != v
        u, v = u[keep] + base, v[keep] + base
        rows += [u, v]
        cols += [v, u]
        base += int(n)
    return _finish(np.concatenate(rows), np.concatenate(cols), sizes, rng, n_feat, n_classes)


def powerlaw_batch(n_graphs, n_lo, n_hi, alpha, seed, n_feat=None, n_classes=2) -> Batch:
    """Configuration-model graphs with Zipf(alpha) degrees clipped to [1, n/2], symmetrised and
    deduplicated (social-network-like hubs)."""
    rng = np.random.default_rng(seed)
    sizes = rng.integers(n_lo, n_hi + 1, size=n_graphs)
    rows, cols = [], []
    base = 0
    for n in sizes:
        deg = np.minimum(rng.zipf(alpha, size=n), max(1, n // 2))
        stubs = np.repeat(np.arange(n), deg)
        rng.shuffle(stubs)
        if stubs.size % 2:
            stubs = stubs[:-1]
        u, v = stubs[0::2], stubs[1::2]
        keep = u != v
        u, v = u[keep] + base, v[keep] + base
        rows += [u, v]
        cols += [v, u]
        base += int(n)
    return _finish(np.concatenate(rows), np.concatenate(cols), sizes, rng, n_feat, n_classes)


def single_graph_batch(n_nodes, nnz, n_feat, n_classes, seed, density=0.01) -> Batch:
    """One undirected graph with exactly ``nnz`` stored entries (``nnz / 2`` distinct edges, no self-loops) and
    sparse binary features (each entry 1 with probability ``density``): the shape of a Planetoid citation graph
    (``rwgnn2/rwgnn_dis_bias6.py:679-931`` loads Cora: 2708 nodes, 10556 entries, 1433 bag-of-words features)."""
    rng = np.random.default_rng(seed)
    want = nnz // 2
    key = np.zeros(0, np.int64)
    while key.size < want:
        u = rng.integers(0, n_nodes, size=2 * want)
        v = rng.integers(0, n_nodes, size=2 * want)
        lo, hi = np.minimum(u, v), np.maximum(u, v)
        k = (lo * n_nodes + hi)[lo != hi]
        cand = np.concatenate([key, k])
        _, first = np.unique(cand, return_index=True)
        key = cand[np.sort(first)][:want]                    # first occurrences, in drawing order
    lo, hi = key // n_nodes, key % n_nodes
    sizes = np.array([n_nodes])
    b = _finish(np.concatenate([lo, hi]), np.concatenate([hi, lo]), sizes, rng, 1, n_classes)
    b.features = (rng.random((n_nodes, n_feat)) < density).astype(np.float32)
    return b


def tu_batch(name, first, count, root=None) -> Batch:
    """Graphs [first, first + count) of a TU-format data set (``rwgnn2/utils.py:11-35``: ``_A.txt`` with 1-based
    ``row, col`` pairs, ``_graph_indicator.txt``, ``_graph_labels.txt``), node degree as the single feature
    (``utils.py:24``), labels renumbered 0..C-1 in sorted order (``LabelEncoder``, ``rwgnn2/main.py:39-40``)."""
    import os
    if root is None:
        root = os.environ.get("RWK_TU_ROOT") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(
            __file__))), "tests", "golden")
    base = os.path.join(root, name, name)
    gi = np.loadtxt(base + "_graph_indicator.txt", dtype=np.int64) - 1
    edges = np.loadtxt(base + "_A.txt", dtype=np.int64, delimiter=",").reshape(-1, 2) - 1
    labels = np.loadtxt(base + "_graph_labels.txt", dtype=np.int64)
    _, y_all = np.unique(labels, return_inverse=True)
    sizes_all = np.bincount(gi)
    n = gi.size
    key = np.unique(edges[:, 0] * n + edges[:, 1])
    r, c = key // n, key % n
    feats = np.bincount(r, minlength=n).astype(np.float32).reshape(n, 1)
    whole = Batch(np.vstack([r, c]), np.ones(key.size, np.float32), feats, gi, y_all.astype(np.int64),
                  sizes_all.astype(np.int64))
    return whole.slice_graphs(first, min(first + count, sizes_all.size))


#: rwgnn2/main.py:17-28 defaults
_DEFAULT_HP = dict(input_dim=1, max_step=2, hidden_graphs=16, size_hidden_graphs=5, hidden_dim=4, penultimate_dim=32,
                   normalize=False, n_classes=2, dropout=0.2)
_PLANETOID_HP = dict(input_dim=64, max_step=3, hidden_graphs=8, size_hidden_graphs=4, hidden_dim=64,
                     penultimate_dim=32, normalize=False, n_classes=7, dropout=0.2)

#: hyper-parameters and generators of the benchmark configurations (BASELINE.json "configs")
CONFIGS = {
    # PROTEINS / D&D-scale batches on one GPU: the configuration the headline metric is quoted on
    "proteins1024": dict(hp=dict(input_dim=1, max_step=2, hidden_graphs=16, size_hidden_graphs=5, hidden_dim=4,
                                 penultimate_dim=32, normalize=False, n_classes=2, dropout=0.2),
                         gen=lambda seed, scale=1: random_batch(1024 * scale, 100, 500, 3.7, seed)),
    # molecule-like stream, MFMA-sized hidden graphs
    "stream_h64": dict(hp=dict(input_dim=16, max_step=2, hidden_graphs=64, size_hidden_graphs=10, hidden_dim=64,
                               penultimate_dim=32, normalize=False, n_classes=2, dropout=0.2),
                       gen=lambda seed, scale=1: random_batch(4096 * scale, 10, 40, 2.2, seed, n_feat=16)),
    # REDDIT-BINARY-like hubs, four propagation steps
    "powerlaw": dict(hp=dict(input_dim=1, max_step=4, hidden_graphs=16, size_hidden_graphs=5, hidden_dim=4,
                             penultimate_dim=32, normalize=False, n_classes=2, dropout=0.2),
                     gen=lambda seed, scale=1: powerlaw_batch(64 * scale, 2000, 4000, 2.1, seed)),
    # one Cora-scale graph with dense features (kept from the first round's measurements)
    "single_graph": dict(hp=dict(input_dim=64, max_step=3, hidden_graphs=8, size_hidden_graphs=4, hidden_dim=64,
                                 penultimate_dim=32, normalize=False, n_classes=7, dropout=0.2),
                         gen=lambda seed, scale=1: random_batch(1, 2708 * scale, 2708 * scale, 3.9, seed,
                                                                n_feat=64, n_classes=7)),
    # D&D-like density of the same batch shape
    "proteins1024_deg5": dict(hp=dict(_DEFAULT_HP), gen=lambda seed, scale=1: random_batch(1024 * scale, 100, 500, 5.0,
                                                                                         seed)),
    # power-law graphs at the wide hidden dimension, and four times as many of them
    "powerlaw_d32": dict(hp=dict(_DEFAULT_HP, max_step=4, hidden_dim=32),
                         gen=lambda seed, scale=1: powerlaw_batch(64 * scale, 2000, 4000, 2.1, seed)),
    "powerlaw256": dict(hp=dict(_DEFAULT_HP, max_step=4),
                        gen=lambda seed, scale=1: powerlaw_batch(256 * scale, 2000, 4000, 2.1, seed)),
    "powerlaw256_d32": dict(hp=dict(_DEFAULT_HP, max_step=4, hidden_dim=32),
                            gen=lambda seed, scale=1: powerlaw_batch(256 * scale, 2000, 4000, 2.1, seed)),
    # Planetoid-scale single graphs (one graph: kernels only, BatchNorm1d cannot train on a batch of one):
    # Cora and PubMed shapes with ~1 % sparse binary features, under the hyper-parameters of
    # rwgnn2/rwgnn_dis_bias6.py:942-946 (8 hidden graphs x 4 nodes, hidden_dim 64, 3 steps) and under main.py's defaults
    "cora": dict(hp=dict(_PLANETOID_HP, input_dim=1433, n_classes=7),
                 gen=lambda seed, scale=1: single_graph_batch(2708, 10556, 1433, 7, seed)),
    "cora_rwnn": dict(hp=dict(_DEFAULT_HP, input_dim=1433, n_classes=7),
                      gen=lambda seed, scale=1: single_graph_batch(2708, 10556, 1433, 7, seed)),
    "pubmed": dict(hp=dict(_PLANETOID_HP, input_dim=500, n_classes=3),
                   gen=lambda seed, scale=1: single_graph_batch(19717, 88648, 500, 3, seed)),
    "pubmed_rwnn": dict(hp=dict(_DEFAULT_HP, input_dim=500, n_classes=3),
                        gen=lambda seed, scale=1: single_graph_batch(19717, 88648, 500, 3, seed)),
    # the reference's own CPU-runnable case: the first batch of 64 MUTAG graphs (1168 nodes, 2590 entries)
    "mutag": dict(hp=dict(_DEFAULT_HP), gen=lambda seed, scale=1: tu_batch("MUTAG", 0, 64 * scale)),
}

def config_batch(name: str, seed: int, scale: int = 1) -> Batch:
    return CONFIGS[name]["gen"](seed, scale)
