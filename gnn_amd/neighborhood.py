"""Two-hop neighbourhood encoding of the distance-biased walk models' fallback path
(``rwgnn2/rwgnn_dis_bias6.py:531-561``, ``_simple_neighborhood_encoding``) on the SpMM kernels.

The reference builds ``A`` as a sparse COO tensor of ones over the edges whose end points are valid (duplicate
edges add up), computes ``A x`` with ``torch.sparse.mm`` and, for ``max_step > 1``, ``A^2 x`` by first forming
``A @ dense(A)`` - an N x N dense matrix - and returns ``(x + A x + 0.5 A^2 x) / 2.5`` (or ``(x + A x) / 2``).
Here ``A^2 x`` is ``A (A x)``: two launches of the CSR SpMM on a structure built once per ``edge_index`` tensor,
no N x N matrix.  Only this encoding is covered; the walk sampling, the GRU encoder and the distance-biased
transition matrix around it (``:109-250``) are not.
"""
from __future__ import annotations

import torch

from .rwconv import _Propagate
from .structure import GraphStructure, IdentityCache

__all__ = ["adjacency_structure", "neighborhood_encoding"]

_CACHE = IdentityCache(256)


def adjacency_structure(edge_index: torch.Tensor, num_nodes: int) -> GraphStructure:
    """CSR of ``A`` (row = ``edge_index[0]``, column = ``edge_index[1]``, value = multiplicity) and of ``A^T``."""
    if edge_index.dim() != 2 or edge_index.shape[0] != 2 or edge_index.dtype != torch.int64:
        raise TypeError("edge_index must be an int64 tensor of shape (2, E)")
    if not edge_index.is_cuda:
        raise RuntimeError("gnn_amd runs on HIP devices only (no CPU fallback)")
    key = (edge_index.data_ptr(), edge_index.shape[1], edge_index._version, int(num_nodes), str(edge_index.device))
    st = _CACHE.get(key)
    if st is not None:
        return st
    dev = edge_index.device
    valid = (edge_index[0] < num_nodes) & (edge_index[1] < num_nodes)          # :537-538
    ei = edge_index[:, valid]
    if ei.shape[1] == 0:
        return None
    a = torch.sparse_coo_tensor(ei, torch.ones(ei.shape[1], device=dev), (num_nodes, num_nodes)).coalesce()
    st = GraphStructure.build(a, torch.zeros(num_nodes, dtype=torch.int64, device=dev))
    _CACHE.put(key, st, (edge_index,))
    return st


def neighborhood_encoding(x_encoded: torch.Tensor, edge_index: torch.Tensor, max_step: int) -> torch.Tensor:
    """``(x + A x + 0.5 A^2 x) / 2.5`` for ``max_step > 1``, ``(x + A x) / 2`` otherwise, ``x`` itself when no edge
    is valid (``rwgnn_dis_bias6.py:540-559``).  Differentiable with respect to ``x_encoded`` (at most 64 channels)."""
    st = adjacency_structure(edge_index, x_encoded.shape[0])
    if st is None:
        return x_encoded
    n1 = _Propagate.apply(x_encoded, st)
    if max_step > 1:
        n2 = _Propagate.apply(n1, st)
        return (x_encoded + n1 + 0.5 * n2) / 2.5
    return (x_encoded + n1) / 2.0
