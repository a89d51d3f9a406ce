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

__all__ = ["load_data", "generate_batches", "DeviceDataset", "BatchPlan"]


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


class _Slot:
    """Preallocated output buffers of one batch (:meth:`DeviceDataset.make_slots`)."""
    __slots__ = ("idx", "vals", "feats", "gi", "y", "crow", "col", "val", "col16", "gid", "seg_ptr", "counts",
                 "counts_f32", "cap", "step", "plan")


class DeviceDataset:
    """All graphs of a data set on the device as one block-diagonal CSR, ready for batch gathers.

    The per-graph sizes stay on the host as well, so the shape of a batch (nodes, stored entries, offsets of its
    graphs, largest row) is known without asking the device: :meth:`batch` is one small upload and ONE launch
    (``rwk_batch_build``) that writes the reference's four tensors and the structure the kernels need (int32 CSR,
    16-bit columns, graph ids, segment pointers, counts), never waits for the device, and so can run on a side
    stream under the previous training step."""

    def __init__(self, adj: Sequence, features: Sequence, y, device):
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
        self._init_arrays(node_ptr, np.concatenate(row_ptr),
                          np.concatenate(cols).astype(np.int32) if cols else np.zeros(0, np.int32),
                          np.concatenate(vals) if vals else np.zeros(0, np.float32), feats,
                          np.asarray(y, dtype=np.int64), device)

    @classmethod
    def from_arrays(cls, node_ptr, row_ptr, col, val, features, y, device) -> "DeviceDataset":
        """From the block-diagonal CSR itself: ``node_ptr`` (G+1) first node of every graph, ``row_ptr`` (N+1),
        ``col`` (data-set node ids, rows column-sorted without duplicates), ``val``, ``features`` (N, F), ``y`` (G)."""
        ds = cls.__new__(cls)
        ds._init_arrays(np.asarray(node_ptr, dtype=np.int64), np.asarray(row_ptr, dtype=np.int64),
                        np.asarray(col, dtype=np.int32), np.asarray(val, dtype=np.float32),
                        np.asarray(features, dtype=np.float32), np.asarray(y, dtype=np.int64), device)
        return ds

    @classmethod
    def from_batch(cls, batch, device) -> "DeviceDataset":
        """The graphs of a ``gnn_amd.synth.Batch`` (row-major sorted COO, block-diagonal) as a data set."""
        n = batch.n_nodes
        row_ptr = np.concatenate([[0], np.cumsum(np.bincount(batch.indices[0], minlength=n))])
        node_ptr = np.concatenate([[0], np.cumsum(batch.sizes)])
        return cls.from_arrays(node_ptr, row_ptr, batch.indices[1], batch.values, batch.features, batch.y, device)

    def _init_arrays(self, node_ptr, row_ptr, col, val, feats, y, device):
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError("gnn_amd assembles batches on HIP devices only (no CPU fallback)")
        if dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        n_all = int(node_ptr[-1])
        if row_ptr.size != n_all + 1 or feats.shape[0] != n_all or y.size != node_ptr.size - 1:
            raise ValueError("inconsistent data set arrays")
        self.device = dev
        self.n_graphs = int(node_ptr.size - 1)
        self.n_feat = int(feats.shape[1])
        # host-side shape of every graph: what batch() needs to size and place a batch without a device read
        self.node_ptr_h = node_ptr
        self.edge_ptr_h = row_ptr[node_ptr]
        deg = np.diff(row_ptr)
        nonempty = np.diff(node_ptr) > 0
        self.graph_max_degree_h = np.zeros(self.n_graphs, dtype=np.int64)
        if n_all and nonempty.any():
            self.graph_max_degree_h[nonempty] = np.maximum.reduceat(deg, node_ptr[:-1][nonempty])
        self.unit_values = bool(np.all(val == np.float32(1.0)))
        self.y_h = y
        self.node_ptr = torch.from_numpy(node_ptr).to(dev)
        self.row_ptr = torch.from_numpy(row_ptr).to(dev)
        self.col = torch.from_numpy(col).to(dev)
        self.val = torch.from_numpy(val).to(dev)
        self.features = torch.from_numpy(np.ascontiguousarray(feats)).to(dev)
        self.y = torch.from_numpy(y).to(dev)
        # A == A^T bit for bit, once per data set (a batch of symmetric graphs is symmetric): one device read here
        self.symmetric = True
        if col.size:
            lib = _lib.load()
            if n_all >= 2 ** 31 - 1 or col.size >= 2 ** 31 - 1:
                raise ValueError("data set too large for 32-bit row pointers")
            with torch.cuda.device(dev):
                flag = torch.zeros(1, dtype=torch.int32, device=dev)
                crow32 = self.row_ptr.to(torch.int32)
                _lib.check(lib.rwk_csr_count_asymmetric(crow32.data_ptr(), self.col.data_ptr(), self.val.data_ptr(),
                                                        n_all, flag.data_ptr(), _lib.current_stream(dev)),
                           "rwk_csr_count_asymmetric")
                self.symmetric = int(flag.item()) == 0

    # ------------------------------------------------------------------------------------
    def batch_shape(self, graph_ids):
        """(nodes, stored entries) of the batch made of these graphs, from the host-side sizes."""
        ids = np.asarray(graph_ids, dtype=np.int64)
        return (int((self.node_ptr_h[ids + 1] - self.node_ptr_h[ids]).sum()),
                int((self.edge_ptr_h[ids + 1] - self.edge_ptr_h[ids]).sum()))

    def make_slots(self, count: int, max_graphs: int, max_nodes: int, max_nnz: int):
        """``count`` sets of preallocated batch buffers for :meth:`batch` (``slot=``): a loop that streams batches
        through them allocates nothing per batch.  The caller orders the reuse of a slot after the last use of the
        batch it held (``gnn_amd.trainer.StreamTrainer`` does)."""
        dev = self.device
        slots = []
        for _ in range(count):
            s = _Slot()
            s.cap = (int(max_graphs), int(max_nodes), int(max_nnz))
            s.idx = torch.empty(2 * max(max_nnz, 1), dtype=torch.int64, device=dev)
            s.vals = torch.empty(max(max_nnz, 1), dtype=torch.float32, device=dev)
            s.feats = torch.empty(max_nodes * max(self.n_feat, 1), dtype=torch.float32, device=dev)
            s.gi = torch.empty(max_nodes, dtype=torch.int64, device=dev)
            s.y = torch.empty(max_graphs, dtype=torch.int64, device=dev)
            s.crow = torch.empty(max_nodes + 1, dtype=torch.int32, device=dev)
            s.col = torch.empty(max(max_nnz, 1), dtype=torch.int32, device=dev)
            s.val = torch.empty(max(max_nnz, 1), dtype=torch.float32, device=dev)
            s.col16 = torch.empty(max(max_nnz, 1), dtype=torch.int16, device=dev)
            s.gid = torch.empty(max_nodes, dtype=torch.int32, device=dev)
            s.seg_ptr = torch.empty(max_graphs + 1, dtype=torch.int32, device=dev)
            s.counts = torch.empty(max_graphs, dtype=torch.int64, device=dev)
            s.counts_f32 = torch.empty(max_graphs, dtype=torch.float32, device=dev)
            slots.append(s)
        return slots

    def plan(self, id_lists) -> "BatchPlan":
        """Host-side shapes and the device-side offset tables of a sequence of batches (lists of graph ids), uploaded
        in one copy: :meth:`batch_planned` then assembles batch ``i`` with one launch and no transfer at all."""
        metas, tables = [], []
        off = 0
        for graph_ids in id_lists:
            ids = np.asarray(graph_ids, dtype=np.int64).reshape(-1)
            b = int(ids.size)
            if b == 0:
                raise ValueError("empty batch")
            if ids.min() < 0 or ids.max() >= self.n_graphs:
                raise IndexError("graph id outside the data set")
            table = np.zeros((4, b + 1), dtype=np.int64)
            np.cumsum(self.node_ptr_h[ids + 1] - self.node_ptr_h[ids], out=table[0, 1:])
            table[1, :b] = self.node_ptr_h[ids]
            np.cumsum(self.edge_ptr_h[ids + 1] - self.edge_ptr_h[ids], out=table[2, 1:])
            table[3, :b] = ids
            n, nnz = int(table[0, b]), int(table[2, b])
            if n == 0:
                raise ValueError("empty batch")
            # |column - first row of the row's 256-row block| must fit int16 for the compact columns
            max_graph = int(np.diff(table[0]).max())
            fits16 = max_graph + 255 <= 32767
            yb = self.y_h[ids]
            metas.append((b, n, nnz, int(self.graph_max_degree_h[ids].max()), fits16, off, int(yb.min()), int(yb.max()),
                          max_graph))
            tables.append(table.reshape(-1))
            off += table.size
        flat = np.concatenate(tables) if len(tables) > 1 else tables[0]
        with torch.cuda.device(self.device):
            dev_tables = torch.from_numpy(flat).pin_memory().to(self.device, non_blocking=True)
        return BatchPlan(dev_tables, metas)

    def batch(self, graph_ids, slot: "_Slot" = None):
        """(adj, features, graph_indicator, y) of the batch made of these graphs, in this order (the loop body of
        ``utils.py:60-83``), with the batch's structure already in the structure cache: ``get_structure(adj,
        graph_indicator)`` - what ``RW_NN.forward`` calls first - finds it without building or reading anything.
        No device-to-host read, one upload (4 x (graphs + 1) int64), one launch on the current stream."""
        return self.batch_planned(self.plan([graph_ids]), 0, slot)

    def batch_planned(self, plan: "BatchPlan", i: int, slot: "_Slot" = None):
        """Batch ``i`` of a :meth:`plan`: one launch (``rwk_batch_build``) on the current stream."""
        dev = self.device
        b, n, nnz, max_degree, fits16, off, y_lo, y_hi, max_graph = plan.metas[i]
        f = self.n_feat
        with _lib.device_guard(dev):
            if slot is None:
                slot = _Slot()
                slot.cap = (b, n, nnz)
                slot.idx = torch.empty(2 * nnz, dtype=torch.int64, device=dev)
                slot.vals = torch.empty(nnz, dtype=torch.float32, device=dev)
                slot.feats = torch.empty(n * f, dtype=torch.float32, device=dev)
                slot.gi = torch.empty(n, dtype=torch.int64, device=dev)
                slot.y = torch.empty(b, dtype=torch.int64, device=dev)
                slot.crow = torch.empty(n + 1, dtype=torch.int32, device=dev)
                slot.col = torch.empty(nnz, dtype=torch.int32, device=dev)
                slot.val = torch.empty(nnz, dtype=torch.float32, device=dev)
                slot.col16 = torch.empty(nnz, dtype=torch.int16, device=dev) if (fits16 and nnz) else None
                slot.gid = torch.empty(n, dtype=torch.int32, device=dev)
                slot.seg_ptr = torch.empty(b + 1, dtype=torch.int32, device=dev)
                slot.counts = torch.empty(b, dtype=torch.int64, device=dev)
                slot.counts_f32 = torch.empty(b, dtype=torch.float32, device=dev)
            self.launch_planned(plan, i, slot, _lib.current_stream(dev))
            return self.wrap_slot(plan, i, slot)

    def launch_planned(self, plan: "BatchPlan", i: int, slot: "_Slot", stream, lean: bool = False):
        """The launch alone: batch ``i`` of the plan into the slot's buffers, queued on ``stream`` (a raw stream
        handle; torch's current stream is left alone).  :meth:`wrap_slot` gives the tensors of the batch; a consumer
        that works on the buffers themselves (``gnn_amd.composite.TrainStep.run_slot``) needs none, and with ``lean``
        the arrays only the tensors are made of stay unwritten (the COO copy of the adjacency, the int64 graph ids,
        the int64 graph sizes, the value array of an unweighted batch with compact columns: 25 of the 45 MB a
        1024-graph batch of proteins-sized graphs writes)."""
        b, n, nnz, max_degree, fits16, off, y_lo, y_hi, max_graph = plan.metas[i]
        cb, cn, ce = slot.cap
        if b > cb or n > cn or nnz > ce:
            raise ValueError(f"batch of {b} graphs, {n} nodes, {nnz} entries exceeds the slot ({cb}, {cn}, {ce})")
        f = self.n_feat
        nz = nnz > 0
        coo = nz and not lean
        compact = fits16 and nz and slot.col16 is not None
        rows_p = slot.idx.data_ptr() if coo else None
        _lib.check(_lib.load().rwk_batch_build(
            self.row_ptr.data_ptr(), _lib.ptr(self.col), _lib.ptr(self.val), self.features.data_ptr(), f,
            self.y.data_ptr(), plan.tables.data_ptr() + 8 * off, b, n, nnz,
            rows_p, rows_p + 8 * nnz if coo else None, slot.vals.data_ptr() if coo else None,
            slot.feats.data_ptr() if f else None, None if lean else slot.gi.data_ptr(), slot.y.data_ptr(),
            slot.crow.data_ptr(), slot.col.data_ptr() if nz else None,
            slot.val.data_ptr() if (nz and not (lean and compact and self.unit_values)) else None,
            slot.col16.data_ptr() if compact else None, slot.gid.data_ptr(),
            slot.seg_ptr.data_ptr(), None if lean else slot.counts.data_ptr(), slot.counts_f32.data_ptr(), stream),
            "rwk_batch_build")

    def wrap_slot(self, plan: "BatchPlan", i: int, slot: "_Slot"):
        """(adj, features, graph_indicator, y) over the slot's buffers after :meth:`launch_planned`, the structure
        registered (host work only)."""
        from .structure import GraphStructure, register_structure
        dev = self.device
        b, n, nnz, max_degree, fits16, off, y_lo, y_hi, max_graph = plan.metas[i]
        f = self.n_feat
        idx = slot.idx[:2 * nnz].view(2, nnz)
        vals = slot.vals[:nnz]
        feats = slot.feats[:n * f].view(n, f)
        gi, yy = slot.gi[:n], slot.y[:b]
        crow, col, val = slot.crow[:n + 1], slot.col[:nnz], slot.val[:nnz]
        col16 = slot.col16[:nnz] if (fits16 and nnz and slot.col16 is not None) else None
        gid, seg_ptr = slot.gid[:n], slot.seg_ptr[:b + 1]
        counts, counts_f32 = slot.counts[:b], slot.counts_f32[:b]
        adj = torch.sparse_coo_tensor(idx, vals, (n, n), check_invariants=False)
        yy._rwk_label_range = (yy._version, y_lo, y_hi)      # known on the host (gnn_amd.head.check_labels)
        if self.symmetric:
            st = GraphStructure.from_parts(dev, n, nnz, b, crow, col, val, col16, self.unit_values, max_degree,
                                           gid, seg_ptr, counts, counts_f32)
            st.plan_tables = plan.tables       # the launch reads them asynchronously
            st.eval_facts = (max_graph, True)  # largest graph; block-diagonal by construction (gnn_amd.eval)
            register_structure(adj, gi, st)
        # an asymmetric data set needs the transposed CSR as well: get_structure builds it on first use
        return adj, feats, gi, yy


class BatchPlan:
    """Offset tables of a sequence of batches on the device and their host-side shapes (``DeviceDataset.plan``)."""
    __slots__ = ("tables", "metas", "_pass_batches")

    def __init__(self, tables, metas):
        self.tables, self.metas = tables, metas
        self._pass_batches = None

    def pass_batches(self, struct):
        """The shapes as an array of ``rwk_pass_batch_t`` (``struct``: its ctypes mirror), made once per plan."""
        if self._pass_batches is None:
            from .structure import spmm_schedule
            arr = (struct * len(self.metas))()
            for c, (b, n, nnz, max_degree, fits16, off, _, _, max_graph) in zip(arr, self.metas):
                c.b, c.n, c.nnz, c.max_graph_nodes, c.table_off = b, n, nnz, max_graph, off
                c.max_degree, c.fits16 = max_degree, int(fits16)
                c.long_threshold = int(spmm_schedule(n, nnz, max_degree)[1] or 0)
            self._pass_batches = arr
        return self._pass_batches

    def __len__(self):
        return len(self.metas)


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
