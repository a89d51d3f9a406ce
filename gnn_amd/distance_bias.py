"""Distance-biased transition operator of the distance-biased walk models as a sparse operator
(``rwgnn2/rwgnn_dis_bias6.py:109-128`` hop distances, ``:222-250`` transition probabilities).

The reference builds, per graph, a dense ``n x n`` distance matrix (hop distance up to ``max_distance``, every other
pair at ``max_distance + 1``) and from it ``P = (1 - c) * rownorm(f(dist), zero diagonal) + c / n`` with
``f(d) = exp(-alpha d)`` ('exp') or ``(d + 1e-6) ** -alpha`` ('levy').  All far pairs share one weight
``w_far = f(max_distance + 1)``, so ``P x`` splits into a sparse product over the ``max_distance``-hop
neighbourhoods plus per-graph column sums:

    (P x)_i = a_i * (w_far * (S_g - x_i) + (Y x)_i) + (c / n_g) * S_g,      S_g = sum of x over graph g,
    Y_ij = f(dist_ij) - w_far  for 1 <= dist_ij <= max_distance,             a_i = (1 - c) / rowsum_i.

The neighbourhood search, the sums and the combination are HIP kernels (``csrc/rwk_khop.hip``); ``Y x`` runs on
the CSR SpMM kernel.  In the reference the matrix only feeds walk *sampling*; this module provides it as the
differentiable linear operator ``x -> P x`` on a whole block-diagonal batch.  HIP devices only.
"""
from __future__ import annotations

import math

import torch

from . import _lib
from .structure import get_structure, spmm_schedule, use_tile_kernel

__all__ = ["DistanceBiasStructure", "distance_bias_structure", "distance_bias_propagate"]


class DistanceBiasStructure:
    """Device-resident pieces of ``P`` for one batch: CSR of ``Y`` (``row_start``, ``col``, ``val``; ``dist`` keeps
    the hop distance of every stored pair), ``row_scale`` (N), ``graph_scale`` (G), ``w_far``."""

    __slots__ = ("base", "n", "max_distance", "row_start", "col", "dist", "val", "level_count", "row_scale",
                 "graph_scale", "w_far", "weights", "max_degree", "row_lanes", "long_threshold", "long_rows")

    def spmm(self, x: torch.Tensor, y: torch.Tensor):
        lib = _lib.load()
        # the launch shape follows the same degree rule as the adjacency product: the tile kernel (row_lanes = 0)
        # for short rows, the packed kernel with a wavefront per very long row once rows hold 32 pairs on average
        nnz = self.col.numel()
        tile = use_tile_kernel(self.n, nnz, self.max_degree, x.shape[1])
        n_long = 0 if (tile or self.long_rows is None) else self.long_rows.numel()
        _lib.check(lib.rwk_spmm_csr(self.row_start.data_ptr(), _lib.ptr(self.col), _lib.ptr(self.val), self.n,
                                    x.shape[1], x.data_ptr(), y.data_ptr(), 0, 0 if tile else self.row_lanes,
                                    _lib.ptr(self.long_rows) if n_long else None, n_long, self.long_threshold,
                                    _lib.current_stream(x.device)), "rwk_spmm_csr")


def _weights(max_distance: int, walk_type: str, alpha: float) -> torch.Tensor:
    d = torch.arange(1, max_distance + 2, dtype=torch.float64)
    if walk_type == "exp":
        return torch.exp(-alpha * d)                                   # :230
    if walk_type == "levy":
        return torch.pow(d + 1e-6, -alpha)                             # :233-234
    raise ValueError("Unsupported types of random walk: %s" % walk_type)   # :237


def distance_bias_structure(adj: torch.Tensor, graph_indicator: torch.Tensor, max_distance: int = 3,
                            walk_type: str = "exp", alpha: float = 1.0, c: float = 0.1) -> DistanceBiasStructure:
    """``adj``: sparse COO (N, N) block-diagonal symmetric adjacency, ``graph_indicator``: sorted int64 graph ids,
    both as ``RW_NN.forward`` takes them.  Defaults: ``max_distance = 3`` (:109), ``alpha = 1`` (:222)."""
    lib = _lib.load()
    base = get_structure(adj, graph_indicator)
    if base.perm is not None:
        raise NotImplementedError("graph_indicator must be sorted (nodes of a graph contiguous)")
    if not base.symmetric:
        raise NotImplementedError("the distance-biased operator needs a symmetric adjacency matrix")
    if not 1 <= int(max_distance) <= 16:
        raise ValueError("max_distance must be in 1..16")
    dev = base.device
    n, md = base.n, int(max_distance)
    st = DistanceBiasStructure()
    st.base, st.n, st.max_distance = base, n, md
    with _lib.device_guard(dev):
        stream = _lib.current_stream(dev)
        max_nodes = int(base.counts.max().item())
        row_count = torch.empty(n, dtype=torch.int32, device=dev)
        st.level_count = torch.empty((n, md), dtype=torch.int32, device=dev)
        csr = base.csr
        _lib.check(lib.rwk_khop_count(csr.crow.data_ptr(), _lib.ptr(csr.col), base.gid.data_ptr(),
                                      base.seg_ptr.data_ptr(), n, max_nodes, md, row_count.data_ptr(),
                                      st.level_count.data_ptr(), stream), "rwk_khop_count")
        starts = torch.zeros(n + 1, dtype=torch.int64, device=dev)
        starts[1:] = torch.cumsum(row_count, 0, dtype=torch.int64)
        nnz = int(starts[-1].item())
        if nnz >= 2 ** 31:
            raise NotImplementedError("more than 2^31 node pairs within max_distance hops")
        st.row_start = starts.to(torch.int32)
        st.max_degree = int(row_count.max().item())
        st.row_lanes, st.long_threshold = spmm_schedule(n, nnz, st.max_degree)
        st.long_rows = torch.nonzero(row_count > st.long_threshold).flatten().to(torch.int32) \
            if st.long_threshold else None
        st.col = torch.empty(max(nnz, 1), dtype=torch.int32, device=dev)[:nnz]
        st.dist = torch.empty(max(nnz, 1), dtype=torch.int32, device=dev)[:nnz]
        if nnz:
            _lib.check(lib.rwk_khop_fill(csr.crow.data_ptr(), _lib.ptr(csr.col), base.gid.data_ptr(),
                                         base.seg_ptr.data_ptr(), n, max_nodes, md, st.row_start.data_ptr(),
                                         st.col.data_ptr(), st.dist.data_ptr(), stream), "rwk_khop_fill")
        f = _weights(md, walk_type, float(alpha)).to(dev)               # f(1) .. f(max_distance + 1), float64
        st.weights = f
        w_far = f[md]
        st.w_far = float(w_far)
        st.val = (f[(st.dist - 1).long()] - w_far).float() if nnz else torch.empty(0, device=dev)
        n_g = base.counts[base.gid.long()].double()
        rowsum = w_far * (n_g - 1.0) + (st.level_count.double() * (f[:md] - w_far)).sum(1)
        rowsum = torch.where(rowsum == 0, torch.ones_like(rowsum), rowsum)      # :240
        st.row_scale = ((1.0 - float(c)) / rowsum).float().contiguous()
        st.graph_scale = (float(c) / base.counts.double()).float().contiguous()  # :243
    return st


class _DistanceBias(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, st: DistanceBiasStructure):
        if not x.is_cuda:
            raise RuntimeError("gnn_amd runs on HIP devices only (no CPU fallback)")
        if x.dim() != 2 or x.shape[0] != st.n or x.dtype != torch.float32:
            raise ValueError("x must be float32 of shape (N, D)")
        ctx.st = st
        return _apply(x.detach().contiguous(), st, None)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        return _apply(g.contiguous().float(), ctx.st, "transpose"), None


def _segment_sum(lib, x, st, stream):
    base = st.base
    out = torch.empty((base.n_graphs, x.shape[1]), dtype=torch.float32, device=x.device)
    _lib.check(lib.rwk_segment_sum(x.data_ptr(), base.seg_ptr.data_ptr(), base.n_graphs, x.shape[1], out.data_ptr(),
                                   stream), "rwk_segment_sum")
    return out


def _apply(x, st, mode):
    """mode None: P x.  "transpose": P^T x (Y is symmetric because hop distances are)."""
    lib = _lib.load()
    dev = x.device
    n, d = x.shape
    if d > 64:
        raise NotImplementedError("at most 64 channels")
    with _lib.device_guard(dev):
        stream = _lib.current_stream(dev)
        s2 = _segment_sum(lib, x, st, stream)
        if mode == "transpose":
            v = torch.empty_like(x)
            _lib.check(lib.rwk_row_scale(x.data_ptr(), st.row_scale.data_ptr(), n, d, v.data_ptr(), stream),
                       "rwk_row_scale")
            s1 = _segment_sum(lib, v, st, stream)
            scale = None
        else:
            v, s1, scale = x, s2, st.row_scale
        y = torch.zeros_like(x) if st.col.numel() == 0 else torch.empty_like(x)
        if st.col.numel():
            st.spmm(v, y)
        out = torch.empty_like(x)
        _lib.check(lib.rwk_distance_bias_combine(v.data_ptr(), y.data_ptr(), s1.data_ptr(), s2.data_ptr(),
                                                 st.base.gid.data_ptr(), _lib.ptr(scale), st.graph_scale.data_ptr(),
                                                 st.w_far, n, d, out.data_ptr(), stream),
                   "rwk_distance_bias_combine")
    return out


def distance_bias_propagate(x: torch.Tensor, st: DistanceBiasStructure) -> torch.Tensor:
    """``P x`` for node features ``x`` (N, D <= 64, float32) of the batch ``st`` was built for; differentiable with
    respect to ``x``."""
    return _DistanceBias.apply(x, st)
