"""TU-format loader and device-side batch construction, drop-in for ``rwgnn2/utils.py:11-85``.

``load_data`` reads ``<root>/<name>/<name>_A.txt`` (1-based ``row, col`` pairs), ``_graph_indicator.txt``,
``_graph_labels.txt`` and optionally ``_node_labels.txt`` and returns what the reference returns: a list of
per-graph scipy CSR adjacency matrices, a list of per-graph feature matrices (node degree, or one-hot node
labels) and the class labels.  The one-hot encoding is built here (sorted distinct labels -> columns, which
is what ``OneHotEncoder`` produces) because the reference's ``OneHotEncoder(sparse=False)`` no longer exists
in current scikit-learn (``utils.py:21``).

``generate_batches`` has the reference's signature and results (lists of sparse COO adjacency, float32
features, int64 ``graph_indicator``, int64 labels, all on ``device``), bit for bit, but assembles every batch
on the device: the whole data set is uploaded once as one block-diagonal CSR, and a batch is a gather of its
graphs' rows (``rwk_batch_gather_coo``) instead of a Python loop filling a scipy LIL matrix.
"""
from __future__ import annotations

import os
from math import ceil
from typing import List, Sequence

import numpy as np
import torch

from . import _lib

__all__ = ["load_data", "generate_batches", "DeviceDataset"]


def load_data(ds_name: str, use_node_labels: bool, root: str = "datasets"):
    """``utils.py:11-35``.  Duplicate edges in ``_A.txt`` add up, as in ``scipy.sparse.csr_matrix``."""
    from scipy.sparse import csr_matrix
    base = os.path.join(root, ds_name, ds_name)
    graph_indicator = np.loadtxt(base + "_graph_indicator.txt", dtype=np.int64)
    _, graph_size = np.unique(graph_indicator, return_counts=True)
    edges = np.loadtxt(base + "_A.txt", dtype=np.int64, delimiter=",").reshape(-1, 2)
    edges = edges - 1
    n = graph_indicator.size
    a = csr_matrix((np.ones(edges.shape[0]), (edges[:, 0], edges[:, 1])), shape=(n, n))
    if use_node_labels:
        labels = np.loadtxt(base + "_node_labels.txt", dtype=np.int64).reshape(-1)
        cats, inv = np.unique(labels, return_inverse=True)
        x = np.zeros((n, cats.size), dtype=np.float64)
        x[np.arange(n), inv] = 1.0
    else:
        x = a.sum(axis=1)
    adj, features = [], []
    idx = 0
    for size in graph_size:
        adj.append(a[idx:idx + size, idx:idx + size])
        features.append(x[idx:idx + size, :])
        idx += size
    class_labels = np.loadtxt(base + "_graph_labels.txt", dtype=np.int64)
    return adj, features, class_labels


class DeviceDataset:
    """All graphs of a data set on the device as one block-diagonal CSR, ready for batch gathers."""

    def __init__(self, adj: Sequence, features: Sequence, y, device):
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError("gnn_amd assembles batches on HIP devices only (no CPU fallback)")
        sizes = np.array([m.shape[0] for m in adj], dtype=np.int64)
        node_ptr = np.concatenate([[0], np.cumsum(sizes)])
        row_ptr = [np.zeros(1, dtype=np.int64)]
        cols, vals = [], []
        nnz = 0
        for g, m in enumerate(adj):
            m = m.tocsr()
            if not m.has_canonical_format:              # column-sorted rows without duplicates, as lil keeps them
                m = m.copy()
                m.sum_duplicates()
            row_ptr.append(m.indptr[1:].astype(np.int64) + nnz)
            cols.append(m.indices.astype(np.int64) + node_ptr[g])
            vals.append(m.data.astype(np.float32))      # utils.py:39 casts the batch to float32
            nnz += int(m.indptr[-1])
        feats = np.vstack([np.asarray(f, dtype=np.float64) for f in features]).astype(np.float32)
        self.device = dev
        self.n_graphs = len(adj)
        self.node_ptr = torch.from_numpy(node_ptr).to(dev)
        self.row_ptr = torch.from_numpy(np.concatenate(row_ptr)).to(dev)
        self.col = torch.from_numpy(np.concatenate(cols).astype(np.int32) if cols else np.zeros(0, np.int32)).to(dev)
        self.val = torch.from_numpy(np.concatenate(vals) if vals else np.zeros(0, np.float32)).to(dev)
        self.features = torch.from_numpy(feats).to(dev)
        self.y = torch.from_numpy(np.asarray(y, dtype=np.int64)).to(dev)

    def batch(self, graph_ids):
        """(adj, features, graph_indicator, y) of the batch made of these graphs, in this order."""
        lib = _lib.load()
        dev = self.device
        with torch.cuda.device(dev):
            ids = torch.as_tensor(np.asarray(graph_ids, dtype=np.int64), device=dev)
            b = ids.numel()
            old_start = self.node_ptr[ids]
            sizes = self.node_ptr[ids + 1] - old_start
            new_start = torch.cumsum(sizes, 0) - sizes
            n = int(sizes.sum().item())
            gi = torch.repeat_interleave(torch.arange(b, device=dev, dtype=torch.int64), sizes, output_size=n)
            delta_g = new_start - old_start
            col_delta = delta_g[gi]
            src_node = torch.arange(n, device=dev, dtype=torch.int64) - col_delta
            deg = self.row_ptr[src_node + 1] - self.row_ptr[src_node]
            out_ptr = torch.cumsum(deg, 0) - deg
            nnz = int(deg.sum().item())
            rows = torch.empty(nnz, dtype=torch.int64, device=dev)
            cols = torch.empty(nnz, dtype=torch.int64, device=dev)
            vals = torch.empty(nnz, dtype=torch.float32, device=dev)
            if n > 0:
                _lib.check(lib.rwk_batch_gather_coo(self.row_ptr.data_ptr(), _lib.ptr(self.col), _lib.ptr(self.val),
                                                    src_node.data_ptr(), col_delta.data_ptr(), out_ptr.data_ptr(), n,
                                                    _lib.ptr(rows), _lib.ptr(cols), _lib.ptr(vals),
                                                    _lib.current_stream(dev)), "rwk_batch_gather_coo")
            adj = torch.sparse_coo_tensor(torch.stack([rows, cols]), vals, (n, n))
            return adj, self.features[src_node], gi, self.y[ids]


def generate_batches(adj, features, y, batch_size, device, shuffle=False):
    """``utils.py:47-85``: same arguments, same four lists, same use of ``np.random.permutation``."""
    n = len(y)
    if shuffle:
        index = np.random.permutation(n)
    else:
        index = np.arange(n, dtype=np.int64)
    ds = DeviceDataset(adj, features, y, device)
    adj_lst: List[torch.Tensor] = []
    features_lst: List[torch.Tensor] = []
    graph_indicator_lst: List[torch.Tensor] = []
    y_lst: List[torch.Tensor] = []
    for i in range(0, n, batch_size):
        a, f, gi, yy = ds.batch(index[i:min(i + batch_size, n)])
        adj_lst.append(a)
        features_lst.append(f)
        graph_indicator_lst.append(gi)
        y_lst.append(yy)
    assert len(adj_lst) == ceil(n / batch_size)
    return adj_lst, features_lst, graph_indicator_lst, y_lst
