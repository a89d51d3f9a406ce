"""Per-batch graph structure on the device: CSR of A and of A^T, sorted-segment metadata, SpMM schedule.

The reference re-derives all of this inside ``torch.spmm`` (COO -> CSR) and ``torch.unique``
(a full sort) on every forward call (``rwgnn2/model.py:32,55``).  Batches are built once per
fold and reused every epoch (``rwgnn2/main.py:70-72``), so here it is computed once per
``(adj, graph_indicator)`` pair and cached by tensor identity.

Integer results (CSR arrays, per-graph node counts, graph ids) are bit-exact with
``adj.coalesce().to_sparse_csr()`` and ``torch.unique(graph_indicator, return_counts=True)``.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Optional

import torch

from . import _lib

__all__ = ["GraphStructure", "get_structure", "register_structure", "clear_structure_cache", "spmm_schedule",
           "use_tile_kernel"]


class _Csr:
    __slots__ = ("crow", "col", "val", "row_lanes", "long_rows", "long_threshold", "max_degree", "col16",
                 "unit_values")

    def __init__(self, crow, col, val):
        self.crow, self.col, self.val = crow, col, val
        self.col16 = None          # int16 column distances for the tile kernel, when they all fit
        self.unit_values = False   # every stored value is exactly 1.0
        self.row_lanes = 1
        self.long_rows = None
        self.long_threshold = 0
        self.max_degree = 0


def _pow2_floor(v: int) -> int:
    p = 1
    while p * 2 <= v:
        p *= 2
    return p


def spmm_schedule(n: int, nnz: int, max_degree: int):
    """Degree-histogram rule for the SpMM launch shape: lanes per row and the long-row cut.

    Mean degree below 8 (molecules, proteins): one lane group per row.  Denser rows share their
    entries over up to 16 lanes.  Rows above ``128 * row_lanes`` entries (power-law hubs) leave
    the packed pass and get a wavefront each.
    """
    mean = nnz / max(n, 1)
    row_lanes = 1 if mean < 8 else min(16, _pow2_floor(int(mean // 4)))
    long_threshold = 128 * row_lanes
    return row_lanes, (long_threshold if max_degree > long_threshold else 0)


def use_tile_kernel(n: int, nnz: int, max_degree: int, d: int) -> bool:
    """Sparse rows go to the tile kernel (``row_lanes = 0``): entry-parallel loads through LDS, hubs summed by
    the whole workgroup.  Dense matrices (mean degree >= 32) and wide operands on very sparse rows (fewer than
    48 stored entries per workgroup of 256/ceil(d/4) rows, and no hubs) stay on the packed kernel."""
    mean = nnz / max(n, 1)
    if mean >= 32:
        return False
    vec = 1
    while vec * 4 < d:
        vec *= 2
    return max_degree > 128 or mean * (256 // vec) >= 48


class GraphStructure:
    """Device-resident structure of one batch.  Build with :meth:`build` (or :func:`get_structure`)."""

    def __init__(self):
        self.device = None
        self.n = 0
        self.nnz = 0
        self.n_graphs = 0
        self.csr: Optional[_Csr] = None
        self.csr_t: Optional[_Csr] = None      # the same object as ``csr`` when A == A^T
        self.symmetric = False
        self.gid = None          # int32 (N): graph id at each sorted position
        self.perm = None         # int32 (N) node index at each sorted position, or None if already sorted
        self.seg_ptr = None      # int32 (G+1)
        self.counts = None       # int64 (G)  == torch.unique(..., return_counts=True)[1]
        self.counts_f32 = None   # float32 (G)
        self.plans = {}          # launch constants per hyper-parameter set (gnn_amd.function._plan)
        self.plan_tables = None  # device buffer the assembling launch reads (DeviceDataset.batch_planned)

    # ------------------------------------------------------------------------------------
    @staticmethod
    def build(adj: torch.Tensor, graph_indicator: torch.Tensor) -> "GraphStructure":
        lib = _lib.load()
        if adj.layout != torch.sparse_coo:
            raise TypeError("adj must be a torch sparse COO tensor (as built by generate_batches)")
        if adj.dim() != 2 or adj.shape[0] != adj.shape[1]:
            raise ValueError("adj must be square (N x N)")
        if not adj.is_cuda:
            raise RuntimeError("gnn_amd runs on HIP devices only: adj is on %s (no CPU fallback)" % adj.device)
        if adj.dtype != torch.float32:
            raise TypeError("adj values must be float32")
        n = adj.shape[0]
        if graph_indicator.dim() != 1 or graph_indicator.shape[0] != n:
            raise ValueError("graph_indicator must have one entry per node")
        if graph_indicator.dtype != torch.int64:
            raise TypeError("graph_indicator must be int64 (torch.LongTensor)")
        if graph_indicator.device != adj.device:
            raise RuntimeError("adj and graph_indicator must be on the same device")
        if n == 0:
            raise ValueError("empty batch")
        dev = adj.device
        st = GraphStructure()
        st.device = dev
        st.n = n
        with torch.cuda.device(dev):
            stream = _lib.current_stream(dev)
            idx = adj._indices()
            vals = adj._values().contiguous()
            rows = idx[0].contiguous()
            cols = idx[1].contiguous()
            gi = graph_indicator.contiguous()
            nnz_in = rows.numel()
            # Every decision that needs a look at the data is launched first and read back ONCE: the inspections of
            # the COO list and of graph_indicator, and - on the assumption they will confirm (row-major sorted COO,
            # as generate_batches builds it) - the CSR, its compact columns, the symmetry test and the largest row.
            #   [0] unsorted COO entries  [1] indices out of range  [4] unsorted ids  [5] max id  [6] negative ids
            #   [7] gaps in sorted ids    [8] columns beyond 16 bits  [9] values != 1  [10] asymmetric entries
            #   [11] largest row
            flags = torch.zeros(16, dtype=torch.int32, device=dev)
            _lib.check(lib.rwk_coo_inspect(_lib.ptr(rows), _lib.ptr(cols), nnz_in, n, flags.data_ptr(), stream),
                       "rwk_coo_inspect")
            _lib.check(lib.rwk_indicator_inspect(gi.data_ptr(), n, flags[4:].data_ptr(), stream),
                       "rwk_indicator_inspect")
            csr = st._csr_from_sorted(lib, rows, cols, vals, n, stream, flags[8:], flags[10:], flags[11:])
            f = flags.cpu().tolist()                      # the one host read-back of a build (once per batch)
            unsorted_coo, out_of_range = f[0], f[1]
            gi_unsorted, gi_max, gi_negative, gi_gaps = f[4], f[5], f[6], f[7]
            if out_of_range:
                raise IndexError("adj has indices outside [0, N)")
            if gi_negative:
                raise IndexError("graph_indicator has negative ids")
            n_graphs = gi_max + 1
            if n_graphs > n or (not gi_unsorted and gi_gaps):
                raise IndexError("graph_indicator ids must cover 0..G-1 without gaps")
            if unsorted_coo:
                # rare path (hand-built inputs): let torch sort and sum duplicates exactly as
                # torch.spmm's own coalesce would, and build again (a second read-back)
                co = adj.coalesce()
                idx = co.indices()
                rows, cols, vals = idx[0].contiguous(), idx[1].contiguous(), co.values().contiguous()
                flags = torch.zeros(16, dtype=torch.int32, device=dev)
                csr = st._csr_from_sorted(lib, rows, cols, vals, n, stream, flags[8:], flags[10:], flags[11:])
                f = flags.cpu().tolist()
            st.csr = csr
            st.nnz = int(rows.numel())
            st._apply_flags(csr, f[8], f[9])
            st.symmetric = f[10] == 0
            st._schedule(csr, f[11])

            if gi_unsorted:
                ids_sorted, order = torch.sort(gi, stable=True)
                st.perm = order.to(torch.int32)
            else:
                ids_sorted = gi
            st.n_graphs = n_graphs
            st.gid = torch.empty(n, dtype=torch.int32, device=dev)
            st.seg_ptr = torch.empty(n_graphs + 1, dtype=torch.int32, device=dev)
            st.counts = torch.empty(n_graphs, dtype=torch.int64, device=dev)
            st.counts_f32 = torch.empty(n_graphs, dtype=torch.float32, device=dev)
            empty = torch.zeros(1, dtype=torch.int32, device=dev)
            _lib.check(lib.rwk_segments_build(ids_sorted.data_ptr(), n, n_graphs, st.gid.data_ptr(),
                                              st.seg_ptr.data_ptr(), st.counts.data_ptr(),
                                              st.counts_f32.data_ptr(), empty.data_ptr(), stream),
                       "rwk_segments_build")
            # sorted ids: "no gaps" was already read with the flags; unsorted ids (rare) ask the segment kernel
            if gi_unsorted and int(empty.item()):
                raise IndexError("graph_indicator ids must cover 0..G-1 without gaps")
            if st.symmetric:
                st.csr_t = st.csr
            else:
                t = torch.sparse_coo_tensor(torch.stack([cols, rows]), vals, (n, n)).coalesce()
                ti = t.indices()
                tflags = torch.zeros(16, dtype=torch.int32, device=dev)
                st.csr_t = st._csr_from_sorted(lib, ti[0].contiguous(), ti[1].contiguous(),
                                               t.values().contiguous(), n, stream, tflags[8:], None, tflags[11:])
                tf = tflags.cpu().tolist()
                st._apply_flags(st.csr_t, tf[8], tf[9])
                st._schedule(st.csr_t, tf[11])
        return st

    @staticmethod
    def from_parts(device, n, nnz, n_graphs, crow, col, val, col16, unit_values, max_degree, gid, seg_ptr, counts,
                   counts_f32) -> "GraphStructure":
        """A structure from arrays the caller already holds (``gnn_amd.data.DeviceDataset.batch`` writes them in its
        one launch and knows the host-side facts from the data set): a symmetric matrix with sorted graph ids.  No
        launch and no device read here, unless the matrix has rows long enough for the packed kernel's long-row
        list."""
        st = GraphStructure()
        st.device, st.n, st.nnz, st.n_graphs = device, int(n), int(nnz), int(n_graphs)
        csr = _Csr(crow, col, val)
        csr.col16 = col16
        csr.unit_values = bool(unit_values)
        st.csr = st.csr_t = csr
        st.symmetric = True
        st.gid, st.seg_ptr, st.counts, st.counts_f32 = gid, seg_ptr, counts, counts_f32
        st._schedule(csr, max_degree)
        return st

    @staticmethod
    def _csr_from_sorted(lib, rows, cols, vals, n, stream, compact_flags, asym_flag, maxdeg_flag) -> _Csr:
        """CSR arrays, compact columns, and (into the given device flag slots) the compact structure's two
        verdicts, the number of asymmetric entries and the largest row.  Nothing is read back here."""
        dev = rows.device
        nnz = rows.numel()
        crow = torch.empty(n + 1, dtype=torch.int32, device=dev)
        col = torch.empty(max(nnz, 1), dtype=torch.int32, device=dev)[:nnz]
        val = torch.empty(max(nnz, 1), dtype=torch.float32, device=dev)[:nnz]
        _lib.check(lib.rwk_csr_from_sorted_coo(_lib.ptr(rows), _lib.ptr(cols), _lib.ptr(vals), nnz, n,
                                               crow.data_ptr(), col.data_ptr(), val.data_ptr(), stream),
                   "rwk_csr_from_sorted_coo")
        csr = _Csr(crow, col, val)
        if nnz:
            csr.col16 = torch.empty(nnz, dtype=torch.int16, device=dev)
            _lib.check(lib.rwk_csr_compact(crow.data_ptr(), col.data_ptr(), val.data_ptr(), n, csr.col16.data_ptr(),
                                           compact_flags.data_ptr(), stream), "rwk_csr_compact")
            if asym_flag is not None:
                _lib.check(lib.rwk_csr_count_asymmetric(crow.data_ptr(), col.data_ptr(), val.data_ptr(), n,
                                                        asym_flag.data_ptr(), stream), "rwk_csr_count_asymmetric")
            torch.amax(crow[1:] - crow[:-1], dim=0, keepdim=True, out=maxdeg_flag[:1])
        return csr

    @staticmethod
    def _apply_flags(csr: _Csr, too_far: int, weighted: int):
        if too_far:
            csr.col16 = None
        csr.unit_values = weighted == 0

    def _schedule(self, csr: _Csr, max_degree: int):
        csr.max_degree = int(max_degree)
        csr.row_lanes, csr.long_threshold = spmm_schedule(self.n, self.nnz, csr.max_degree)
        if csr.long_threshold:
            deg = csr.crow[1:] - csr.crow[:-1]
            csr.long_rows = torch.nonzero(deg > csr.long_threshold).flatten().to(torch.int32)

    # ------------------------------------------------------------------------------------
    def spmm(self, x: torch.Tensor, y: torch.Tensor, transpose: bool = False, accumulate: bool = False,
             kernel: Optional[str] = None, reduce_job=None):
        """y = A x (or A^T x), optionally accumulated into y.  x, y: contiguous (N, D) float32.
        ``kernel``: "tile", "tile32" (tile kernel on the int32 columns and the value array) or "packed" to force
        a schedule (tests); default: the degree-histogram rule, compact structure when it exists.
        ``reduce_job`` = (partials (P, V), out32 (32, V)), P > 32: also run the first level of the deterministic
        row reduction of ``partials`` (``rwk_reduce_rows_to32``), in the SpMM's own launch when the tile kernel
        serves the product."""
        lib = _lib.load()
        csr = self.csr_t if transpose else self.csr
        stream = _lib.current_stream(self.device)
        tile = use_tile_kernel(self.n, self.nnz, csr.max_degree, x.shape[1]) if kernel is None \
            else kernel in ("tile", "tile32")
        compact = tile and csr.col16 is not None and kernel != "tile32"
        if tile and reduce_job is not None:
            red_in, red_out = reduce_job
            _lib.check(lib.rwk_spmm_tile_reduce_rows(
                csr.crow.data_ptr(), _lib.ptr(csr.col), csr.col16.data_ptr() if compact else None,
                None if (compact and csr.unit_values) else _lib.ptr(csr.val), self.n, x.shape[1], x.data_ptr(),
                y.data_ptr(), int(accumulate), red_in.data_ptr(), red_in.shape[0], red_in.shape[1],
                red_out.data_ptr(), stream), "rwk_spmm_tile_reduce_rows")
            return
        if compact:
            _lib.check(lib.rwk_spmm_csr_compact(csr.crow.data_ptr(), csr.col16.data_ptr(),
                                                None if csr.unit_values else csr.val.data_ptr(), self.n, x.shape[1],
                                                x.data_ptr(), y.data_ptr(), int(accumulate), stream),
                       "rwk_spmm_csr_compact")
        else:
            n_long = 0 if (tile or csr.long_rows is None) else csr.long_rows.numel()
            _lib.check(lib.rwk_spmm_csr(csr.crow.data_ptr(), _lib.ptr(csr.col), _lib.ptr(csr.val), self.n,
                                        x.shape[1], x.data_ptr(), y.data_ptr(), int(accumulate),
                                        0 if tile else csr.row_lanes,
                                        _lib.ptr(csr.long_rows), n_long, csr.long_threshold, stream), "rwk_spmm_csr")
        if reduce_job is not None:
            red_in, red_out = reduce_job
            _lib.check(lib.rwk_reduce_rows_to32(red_in.data_ptr(), red_in.shape[0], red_in.shape[1],
                                                red_out.data_ptr(), stream), "rwk_reduce_rows_to32")

    def spmm_bytes(self, d: int) -> int:
        """Algorithmic HBM bytes of one SpMM launch: values + columns + row pointers + x + y."""
        return self.nnz * 8 + (self.n + 1) * 4 + 2 * self.n * d * 4

    def spmm_bytes_moved(self, d: int, transpose: bool = False) -> int:
        """Bytes the scheduled kernel reads and writes: as :meth:`spmm_bytes`, with 2 bytes per stored entry for
        16-bit column distances and none for values that are all 1.0 when the tile kernel runs on them."""
        csr = self.csr_t if transpose else self.csr
        per_entry = 8
        if csr.col16 is not None and use_tile_kernel(self.n, self.nnz, csr.max_degree, d):
            per_entry = 2 + (0 if csr.unit_values else 4)
        return self.nnz * per_entry + (self.n + 1) * 4 + 2 * self.n * d * 4


# ----------------------------------------------------------------------------------------
# cache keyed by tensor identity
# ----------------------------------------------------------------------------------------
class IdentityCache:
    """Structures keyed by the identity (data pointers, versions) of the tensors they were built from.

    An entry lives as long as those tensors do: it holds no reference to them, and a finalizer on each of them drops
    the entry when the tensor object is collected - so the batches of a finished cross-validation fold
    (``rwgnn2/main.py:47-72`` builds new ones per fold) release their CSR arrays with them, and a data pointer cannot
    be met again by another batch while its entry exists.  ``capacity`` bounds the number of entries as a backstop."""

    def __init__(self, capacity: int):
        self.capacity = capacity
        self.entries: "OrderedDict[tuple, object]" = OrderedDict()

    def get(self, key):
        st = self.entries.get(key)
        if st is not None:
            self.entries.move_to_end(key)
        return st

    def put(self, key, st, owners):
        import weakref
        self.entries[key] = st
        self.entries.move_to_end(key)
        for t in owners:
            if t is not None:
                weakref.finalize(t, self._drop, key, id(st))
        while len(self.entries) > self.capacity:
            self.entries.popitem(last=False)

    def _drop(self, key, st_id):
        # only the entry this owner registered: buffers that are reused (batch slots) meet the same key again, and a
        # late finalizer of the previous batch's tensors must not remove the new batch's entry
        st = self.entries.get(key)
        if st is not None and id(st) == st_id:
            del self.entries[key]

    def clear(self):
        self.entries.clear()

    def __len__(self):
        return len(self.entries)


_CACHE = IdentityCache(512)


def _key(adj, graph_indicator):
    idx, val = adj._indices(), adj._values()
    return (idx.data_ptr(), val.data_ptr(), graph_indicator.data_ptr(), idx.shape[1], adj.shape[0],
            idx._version, val._version, graph_indicator._version, str(adj.device))


def get_structure(adj, graph_indicator) -> GraphStructure:
    """Cached :meth:`GraphStructure.build`; the entry is dropped when ``adj`` or ``graph_indicator`` is collected."""
    if isinstance(adj, GraphStructure):
        return adj
    key = _key(adj, graph_indicator)
    st = _CACHE.get(key)
    if st is not None:
        return st
    st = GraphStructure.build(adj, graph_indicator)
    _CACHE.put(key, st, (adj, graph_indicator))
    return st


def register_structure(adj, graph_indicator, st: GraphStructure):
    """Enter a structure built elsewhere (``DeviceDataset.batch``) for this ``(adj, graph_indicator)`` pair; it
    replaces any entry with the same key (reused batch buffers)."""
    _CACHE.put(_key(adj, graph_indicator), st, (adj, graph_indicator))


def clear_structure_cache():
    _CACHE.clear()
